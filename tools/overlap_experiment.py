#!/usr/bin/env python3
"""Experiment: one batch of reads coded as chunks on several contexts (streams) instead of one launch sequence.

    python tools/overlap_experiment.py [--reads 32768] [--chunks 1024,2048,4096,8192] [--contexts 1,2,4]

Question it answers: does a chunk's svb intermediate stay in the 256 MB Infinity Cache when the
chunks are small, and do chunks on different streams fill each other's launch tails?  Prints one JSON line per
(chunk, contexts) pair: encode+decode MB/s of raw int16 against the single-launch-sequence baseline."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=32768)
    ap.add_argument("--chunks", default="1024,2048,4096,8192")
    ap.add_argument("--contexts", default="1,2,4")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    from vbz_compression_amd import batch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    maxctx = max(int(x) for x in args.contexts.split(","))
    codecs = [batch.GpuCodec(0) for _ in range(maxctx)]
    c0 = codecs[0]
    opts = c0.options(True, 2, 1, 1)
    n = args.reads
    with torch.cuda.stream(c0.stream):
        lens = c0.synth_lengths(5, 0, n)
        sizes = lens.to(torch.int64) * 2
        off, total = batch.layout(sizes.cpu(), 64)
        caps = bench.max_compressed_sizes(sizes.cpu(), 1)
        coff, ctotal = batch.layout(caps, 64)
        raw = torch.empty(total, dtype=torch.uint8, device=dev)
        offd = off.to(dev)
        c0.synth_signal(5, 0, raw, offd, lens)
        comp = torch.empty(ctotal, dtype=torch.uint8, device=dev)
        back = torch.empty(total, dtype=torch.uint8, device=dev)
        size32 = sizes.to(torch.int32)
        cap32 = caps.to(torch.int32).to(dev)
        coffd = coff.to(dev)
        csize = torch.zeros(n, dtype=torch.int32, device=dev)
        res = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    raw_bytes = int(sizes.sum())

    def chunk_views(a, b):
        o0, o1 = int(off[a]), (int(off[b]) if b < n else total)
        c0_, c1_ = int(coff[a]), (int(coff[b]) if b < n else ctotal)
        return dict(raw=raw[o0:o1], off=offd[a:b] - o0, size=size32[a:b], comp=comp[c0_:c1_], coff=coffd[a:b] - c0_, cap=cap32[a:b],
                    csize=csize[a:b], back=back[o0:o1], res=res[a:b])

    def run(chunk, nctx):
        views = [chunk_views(a, min(a + chunk, n)) for a in range(0, n, chunk)]
        torch.cuda.synchronize()

        def one_pass():
            for i, v in enumerate(views):
                c = codecs[i % nctx]
                with torch.cuda.stream(c.stream):
                    c.compress(v["raw"], v["off"], v["size"], v["comp"], v["coff"], v["cap"], v["csize"], opts)
                    c.decompress(v["comp"], v["coff"], v["csize"], v["back"], v["off"], v["size"], v["res"], opts)

        one_pass()
        torch.cuda.synchronize()
        assert bool((res == size32).all()) and torch.equal(raw, back)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_pass()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    base = run(n, 1)
    print(json.dumps({"chunk": n, "contexts": 1, "ms": round(base * 1e3, 3), "MBps": round(raw_bytes / base / 1e6, 1), "note": "one launch sequence"}), flush=True)
    for chunk in [int(x) for x in args.chunks.split(",")]:
        for nctx in [int(x) for x in args.contexts.split(",")]:
            t = run(chunk, nctx)
            print(json.dumps({"chunk": chunk, "contexts": nctx, "ms": round(t * 1e3, 3), "MBps": round(raw_bytes / t / 1e6, 1),
                              "vs_single": round(base / t, 3)}), flush=True)


if __name__ == "__main__":
    main()
