#!/usr/bin/env python3
"""The reference's own measurement matrix (vbz/perf/vbz_perf.cpp:113-171: compress / decompress x {sequence, signal} x int8 / int16 / int32 x zstd {1, 0},
zig-zag on, default version) on the MI355X next to the reference path restated on this box's CPU (the oracle, ONE thread -- the reference's
benchmark is single-threaded).

  sequence   SequenceGenerator (test_data_generator.h:12-23): ONE buffer of 1 MB of iota -- one buffer per call, as the reference times it
             (a latency figure on a GPU), and 256 of them per call
  signal     SignalGenerator (:28-74): 100 MB of reads of 30 000 - 200 000 values that cycle the 15 643-sample read of vbz/test/test_data.h,
             handed over as ONE batch (inputs resident in HBM)

    python tools/perf_matrix.py [--md profiles/r06_perf_matrix.md] [--mb 100]

MB/s are of raw integer bytes (vbz_perf.cpp:45-46: SetBytesProcessed(items x int_size))."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default="")
    ap.add_argument("--mb", type=int, default=100, help="megabytes of the signal set (the reference's byte_target)")
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import ratio_sweep
    from vbz_compression_amd import batch

    c = batch.GpuCodec(0)
    dev = c.device
    L = c.L
    rng = np.random.default_rng(5)
    t = ratio_sweep.template()
    rows = []
    for dt in (np.int8, np.int16, np.int32):
        isz = np.dtype(dt).itemsize
        seq = np.arange(1000 * 1000 // isz, dtype=np.int64).astype(dt)
        sig, total = [], 0
        while total < args.mb * 1000 * 1000:
            n = min(int(rng.integers(30000, 200001)), (args.mb * 1000 * 1000 - total) // isz)
            if n <= 0:
                break
            sig.append(np.resize(t, n).astype(dt))
            total += n * isz
        for level in (1, 0):
            opts = c.options(True, isz, level, 0)
            oo = O.options(True, isz, level, 0)
            for name, arrays in (("sequence, 1 buffer per call", [seq]), ("sequence, 256 buffers per call", [seq] * 256), ("signal, %d MB in one call" % args.mb, sig)):
                sizes = torch.tensor([a.nbytes for a in arrays], dtype=torch.int64)
                off, tot = batch.layout(sizes, 64)
                caps = torch.tensor([L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.tolist()], dtype=torch.int64)
                coff, ctot = batch.layout(caps, 64)
                arena = np.zeros(tot + 64, np.uint8)
                for a, o in zip(arrays, off.tolist()):
                    arena[o : o + a.nbytes] = np.frombuffer(a.tobytes(), np.uint8)
                src = torch.from_numpy(arena).to(dev)
                comp = torch.zeros(ctot + 64, dtype=torch.uint8, device=dev)
                back = torch.zeros(tot + 64, dtype=torch.uint8, device=dev)
                csize = torch.zeros(len(arrays), dtype=torch.int32, device=dev)
                res = torch.zeros(len(arrays), dtype=torch.int32, device=dev)
                offd, coffd, s32, c32 = off.to(dev), coff.to(dev), sizes.to(torch.int32).to(dev), caps.to(torch.int32).to(dev)
                raw = int(sizes.sum())
                reps = 20 if raw < 50e6 else 5
                with torch.cuda.stream(c.stream):
                    c.compress(src, offd, s32, comp, coffd, c32, csize, opts)
                    c.decompress(comp, coffd, csize, back, offd, s32, res, opts)
                    torch.cuda.synchronize()
                    assert bool((res == s32).all()) and torch.equal(src[:tot], back[:tot]), name
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record()
                    for _ in range(reps):
                        c.compress(src, offd, s32, comp, coffd, c32, csize, opts)
                    e[1].record()
                    for _ in range(reps):
                        c.decompress(comp, coffd, csize, back, offd, s32, res, opts)
                    e[2].record()
                torch.cuda.synchronize()
                te, td = e[0].elapsed_time(e[1]) / reps, e[1].elapsed_time(e[2]) / reps
                cbytes = int(csize.to(torch.int64).sum())
                # the reference path on ONE CPU thread: a bounded sample of the same buffers
                sample = arrays[: max(1, min(len(arrays), int(20e6 // max(arrays[0].nbytes, 1))))]
                t0 = time.perf_counter()
                frames = [O.compress(a, oo) for a in sample]
                t1 = time.perf_counter()
                for a, f in zip(sample, frames):
                    O.decompress(f, a.nbytes, oo)
                t2 = time.perf_counter()
                sraw = sum(a.nbytes for a in sample)
                rows.append((np.dtype(dt).name, level, name, raw / te / 1e3, raw / td / 1e3, raw / cbytes, sraw / (t1 - t0) / 1e6, sraw / (t2 - t1) / 1e6,
                             sraw / sum(len(f) for f in frames)))
    lines = ["# The reference's measurement matrix (vbz/perf/vbz_perf.cpp:113-171) on one MI355X (tools/perf_matrix.py)", "",
             "zig-zag on, version 0; MB/s of raw integer bytes; GPU: batched entry points, inputs resident in HBM, HIP events; CPU: the reference path restated (oracle + libzstd "
             + (O.lib().vbo_zstd_version() or b"?").decode() + "), ONE thread, on this box.", "",
             "| type | zstd level | input | GPU compress MB/s | GPU decompress MB/s | ratio | CPU compress MB/s | CPU decompress MB/s | CPU ratio |", "|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | %d | %s | %.0f | %.0f | %.2f | %.0f | %.0f | %.2f |" % r)
    text = "\n".join(lines) + "\n"
    print(text)
    if args.md:
        with open(args.md, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
