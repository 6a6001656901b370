#!/usr/bin/env python3
"""A/B of two (or more) builds of libvbz_hip.so inside ONE process, alternating call by call, so that every build sees the
same clocks and temperatures (consecutive bench.py runs drift by several per cent while the board warms up: the first run of
a call is the fastest).  Per-kernel average durations from each library's own HIP events:

    python tools/ab_libs.py xlibs/base.so xlibs/new.so [--reads 65536] [--rounds 12]

Each library encodes and decodes its own frames (a build may write trailers the other does not read).  The context that is
created SECOND runs its memory-bound kernels up to 5 % slower whatever its code (where its scratch arena lands in HBM): run
both orders (`a b` and `b a`) and compare like positions."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbz_compression_amd import _lib, batch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--reads", type=int, default=65536)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--level", type=int, default=1)
args = ap.parse_args()

codecs = []
for path in args.libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    codecs.append(batch.GpuCodec(0))
c0 = codecs[0]
torch.cuda.set_stream(c0.stream)
opts = c0.options(True, 2, args.level, 1)
n = args.reads
lens = c0.synth_lengths(5, 0, n)
sizes = lens.to(torch.int64) * 2
off, total = batch.layout(sizes.cpu(), 64)
raw = torch.empty(total, dtype=torch.uint8, device="cuda")
off = off.cuda()
c0.synth_signal(5, 0, raw, off, lens)
s32 = sizes.to(torch.int32)
caps = torch.tensor([c0.L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
coff, ctotal = batch.layout(caps, 64)
coff = coff.cuda()
cap32 = caps.to(torch.int32).cuda()
comp = torch.empty(ctotal, dtype=torch.uint8, device="cuda")
back = torch.empty_like(raw)
state = [(torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")) for _ in codecs]
torch.cuda.synchronize()


def step(c, st):
    cs, res = st
    with torch.cuda.stream(c.stream):
        c.compress(raw, off, s32, comp, coff, cap32, cs, opts)
        c.decompress(comp, coff, cs, back, off, s32, res, opts)
    c.synchronize()


for c, st in zip(codecs, state):
    step(c, st)
    if not torch.equal(raw, back): print("(round trip differs:", os.path.basename(c.L._name), ")")
    back.zero_()
for c in codecs:
    c.profile_reset()
    c.profile(True)
for _ in range(args.rounds):
    for c, st in zip(codecs, state):
        step(c, st)
for path, c, st in zip(args.libs, codecs, state):
    c.profile(False)
    per = {k: round(v[1] / max(v[0], 1), 3) for k, v in c.profile_read().items() if "zstd" in k or "svb" in k}
    print("%-24s ratio %.4f  sum %.3f ms  %s" % (os.path.basename(path), float(sizes.sum()) / float(st[0].to(torch.int64).sum()), sum(per.values()), per))
