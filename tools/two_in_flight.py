#!/usr/bin/env python3
"""Two batches in flight: two contexts of the library, each with a stream of its own, code alternate batches of the headline workload
(encode then decode, resident inputs) without waiting for each other -- the kernels of one batch's latency-bound launches run beside
the other batch's memory-bound ones.  Prints the aggregate rate next to one context doing the same number of steps alone.

    python tools/two_in_flight.py [--reads 32768] [--steps 12] [--contexts 2]

An experiment, not the benchmark: bench.py's step is one batch at a time on one stream (its per-kernel times and its roofline figure
are sums over that stream), and that is what its `value` stays."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vbz_compression_amd import batch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=32768)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--contexts", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def make(codec, first, n):
    lens = codec.synth_lengths(5, first, n)
    sizes = lens.to(torch.int64) * 2
    off, total = batch.layout(sizes.cpu(), 64)
    caps = bench.max_compressed_sizes(sizes.cpu(), 1)
    coff, ctotal = batch.layout(caps, 64)
    raw = torch.zeros(total, dtype=torch.uint8, device=dev)
    off = off.to(dev)
    codec.synth_signal(5, first, raw, off, lens)
    return dict(n=n, raw=raw, off=off, size=sizes.to(torch.int32).to(dev), coff=coff.to(dev), cap=caps.to(torch.int32).to(dev),
                csize=torch.zeros(n, dtype=torch.int32, device=dev), raw_bytes=int(sizes.sum()), total=total,
                comp=torch.empty(ctotal, dtype=torch.uint8, device=dev), back=torch.zeros(total, dtype=torch.uint8, device=dev),
                res=torch.zeros(n, dtype=torch.int32, device=dev))


codecs = [batch.GpuCodec(0) for _ in range(args.contexts)]
opts = codecs[0].options(True, 2, 1, 1)
work = []
for k, c in enumerate(codecs):
    with torch.cuda.stream(c.stream):
        work.append([make(c, (2 * k + j) * args.reads, args.reads) for j in range(2)])
torch.cuda.synchronize()


def step(c, B):
    with torch.cuda.stream(c.stream):
        c.compress(B["raw"], B["off"], B["size"], B["comp"], B["coff"], B["cap"], B["csize"], opts)
        c.decompress(B["comp"], B["coff"], B["csize"], B["back"], B["off"], B["size"], B["res"], opts)


for k, c in enumerate(codecs):
    for B in work[k]:
        step(c, B)
torch.cuda.synchronize()
for k, c in enumerate(codecs):
    for B in work[k]:
        assert bool((B["res"] == B["size"]).all()) and torch.equal(B["raw"], B["back"]), "round trip"


def run(active):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for i in range(args.steps):
        for k in active:
            step(codecs[k], work[k][i & 1])
            done += work[k][i & 1]["raw_bytes"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return done / dt / 1e9, dt * 1e3 / (args.steps * len(active))


for _ in range(2):
    alone = run([0])
    both = run(list(range(args.contexts)))
    print("reads per batch %d: one context alone %.1f GB/s (%.2f ms per step); %d contexts in flight %.1f GB/s aggregate (%.2f ms per step)"
          % (args.reads, alone[0], alone[1], args.contexts, both[0], both[1]), flush=True)
