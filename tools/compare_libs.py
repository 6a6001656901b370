#!/usr/bin/env python3
"""Do two builds of libvbz_hip.so write the same bytes?  Each build runs in a process of its own (VBZ_HIP_LIB) over the same inputs --
synthetic signal of the benchmark's generator, template-cycling reads, noise, constants, short and ragged reads -- and prints one
sha256 per case of everything it wrote; the parent compares.  For changes that must not change a frame (a kernel restructured,
work moved from one launch to another):

    python tools/compare_libs.py xlibs/base.so vbz_compression_amd/lib/libvbz_hip.so [--reads 4096] [--env VBZ_HIP_STAGED_ENCODE=0]

--env applies to the SECOND library only (the same library twice with a knob is a comparison too).  Exit code 1 on any difference."""
import argparse
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, hashlib, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import oracle_lib as O, gpu_util as G
from vbz_compression_amd import _lib, batch
n_big = int(sys.argv[1])
def digest(frames):
    h = hashlib.sha256()
    for f in frames:
        h.update(b"E%%d" %% f if isinstance(f, int) else f.tobytes())
        h.update(b"|")
    return h.hexdigest(), sum(0 if isinstance(f, int) else len(f) for f in frames)
rng = np.random.default_rng(11)
cases = {}
cases["signal_ragged"] = [O.synth_signal(5, i, n) for i, n in enumerate([0, 1, 17, 255, 1023, 1639, 1640, 1641, 2047, 2048, 2049, 4096, 8191, 8192, 30000, 32767, 32768, 32769, 65536, 100000, 131071, 250000, 524288, 600000])]
t = O.synth_signal(5, 99, 15643)
cases["cycled"] = [np.resize(t, n) for n in (30000, 100000, 200000)]
cases["noise"] = [rng.integers(-32768, 32767, n, dtype=np.int16) for n in (5000, 50000, 100000)]
cases["narrow"] = [rng.integers(-3, 4, n).astype(np.int16).cumsum().astype(np.int16) for n in (5000, 50000, 100000)]
cases["constant"] = [np.full(n, 7, np.int16) for n in (3000, 40000)] + [np.arange(0, n, dtype=np.int16) for n in (1000, 40000)]
cases["steps"] = [np.repeat(rng.integers(-2000, 2000, 400).astype(np.int16), 250)[:n] for n in (20000, 100000)]
out = {}
for name, reads in cases.items():
    for lvl, sized in ((1, False), (1, True), (4, False)):
        opts = _lib.CompressionOptions(True, 2, lvl, 1)
        frames = G.compress(reads, opts, sized=sized)
        out["%%s/l%%d%%s" %% (name, lvl, "s" if sized else "")] = digest(frames)
        back = G.decompress(frames, [a.nbytes for a in reads], opts, sized=sized)
        for a, bk in zip(reads, back):
            assert not isinstance(bk, int) and bk.tobytes() == a.tobytes(), name
for size, zz in ((4, False), (4, True), (1, True), (2, False)):
    dt = {1: np.int8, 2: np.int16, 4: np.uint32}[size]
    reads = [O.synth_u32(5, i, n).astype(dt) if size == 4 else O.synth_signal(5, i, n).astype(dt) for i, n in enumerate([3000, 20000, 70000])]
    opts = _lib.CompressionOptions(zz, size, 1, 0)
    out["other/%%d%%s" %% (size, "z" if zz else "")] = digest(G.compress(reads, opts))
# a resident batch of the benchmark's generator (one-wavefront path at scale)
c = G.codec(); L = c.L
opts = c.options(True, 2, 1, 1)
lens = c.synth_lengths(5, 0, n_big); sizes = lens.to(torch.int64) * 2
off, total = batch.layout(sizes.cpu(), 64); raw = torch.empty(total, dtype=torch.uint8, device="cuda"); off = off.cuda()
c.synth_signal(5, 0, raw, off, lens); s32 = sizes.to(torch.int32)
caps = torch.tensor([L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
coff, ctotal = batch.layout(caps, 64); comp = torch.zeros(ctotal, dtype=torch.uint8, device="cuda"); coff = coff.cuda(); cap32 = caps.to(torch.int32).cuda()
cs = torch.zeros(n_big, dtype=torch.int32, device="cuda")
c.compress(raw, off, s32, comp, coff, cap32, cs, opts); torch.cuda.synchronize()
h = hashlib.sha256(); host = comp.cpu().numpy(); tot = 0
for o, k in zip(coff.cpu().tolist(), cs.cpu().tolist()):
    h.update(host[o:o + k].tobytes()); tot += k
out["bench_batch/%%d" %% n_big] = (h.hexdigest(), tot)
back = torch.empty_like(raw); res = torch.zeros(n_big, dtype=torch.int32, device="cuda")
c.decompress(comp, coff, cs, back, off, s32, res, opts); torch.cuda.synchronize()
assert torch.equal(raw, back)
print("VERSION", L.vbz_gpu_version().decode())
for k, (d, nb) in out.items():
    print("CASE", k, d, nb)
""" % (ROOT, ROOT)


def run(lib, reads, extra_env):
    env = dict(os.environ, VBZ_HIP_LIB=os.path.abspath(lib))
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", CHILD, str(reads)], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:] + r.stderr[-3000:])
        raise SystemExit("compare_libs: %s failed" % lib)
    cases = {}
    version = "?"
    for ln in r.stdout.splitlines():
        if ln.startswith("CASE"):
            _, k, d, nb = ln.split()
            cases[k] = (d, int(nb))
        elif ln.startswith("VERSION"):
            version = ln[8:]
    return version, cases


ap = argparse.ArgumentParser()
ap.add_argument("a")
ap.add_argument("b")
ap.add_argument("--reads", type=int, default=4096)
ap.add_argument("--env", action="append", default=[])
args = ap.parse_args()
extra = dict(e.split("=", 1) for e in args.env)
va, ca = run(args.a, args.reads, {})
vb, cb = run(args.b, args.reads, extra)
print("A: %s (%s)\nB: %s (%s) %s" % (args.a, va, args.b, vb, extra or ""))
bad = 0
for k in ca:
    same = ca[k][0] == cb.get(k, ("", 0))[0]
    bad += not same
    print("%-24s %s  %10d / %10d bytes" % (k, "same" if same else "DIFFERENT", ca[k][1], cb.get(k, ("", 0))[1]))
print("compare_libs:", "identical output" if not bad else "%d case(s) differ" % bad)
sys.exit(1 if bad else 0)
