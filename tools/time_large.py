#!/usr/bin/env python3
"""Time the large-read path through the batched entry points with n_reads = 1:

    python tools/time_large.py            # config 4 (one 10 M-element uint32 buffer) and config 1 (one 400 k-sample int16 read)

Prints one JSON line per case: GB/s of raw bytes each way (device-resident, HIP events), and the same through the
single-buffer host API (vbz_compress / vbz_decompress: PCIe copies and synchronisation included).  VBZ_HIP_SEGMENTED=0
times the one-workgroup-per-read kernels on the same input."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from vbz_compression_amd import _lib, batch, vbz

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = batch.GpuCodec(0)
    L = c.L
    torch.cuda.set_stream(c.stream)
    cases = [("config4: one 10M-element uint32 buffer, no zig-zag, level 3", 4, False, 3, 0, 10_000_000, "u32"),
             ("config1: one 400k-sample int16 read, zig-zag, level 1", 2, True, 1, 1, 400_000, "i16"),
             ("one 4M-sample int16 read", 2, True, 1, 1, 4_000_000, "i16")]
    if len(sys.argv) > 1:
        cases = [cases[int(k)] for k in sys.argv[1].split(",")]
    for name, size, zz, level, ver, count, kind in cases:
        opts = c.options(zz, size, level, ver)
        nbytes = count * size
        off = torch.zeros(1, dtype=torch.int64, device=dev)
        lens = torch.tensor([count], dtype=torch.int32, device=dev)
        raw = torch.zeros(nbytes + 64, dtype=torch.uint8, device=dev)
        if kind == "u32":
            c.synth_u32(5, 3, raw, off, lens)
        else:
            c.synth_signal(5, 0, raw, off, lens)
        cap = L.vbz_max_compressed_size(nbytes, ctypes.byref(opts))
        comp = torch.zeros(cap + 64, dtype=torch.uint8, device=dev)
        back = torch.zeros_like(raw)
        size32 = torch.tensor([nbytes], dtype=torch.int32, device=dev)
        cap32 = torch.tensor([cap], dtype=torch.int64).to(torch.int32).to(dev)
        csize = torch.zeros(1, dtype=torch.int32, device=dev)
        res = torch.zeros(1, dtype=torch.int32, device=dev)

        def enc():
            c.compress(raw, off, size32, comp, off, cap32, csize, opts)

        def dec():
            c.decompress(comp, off, csize, back, off, size32, res, opts)

        enc(); dec()
        torch.cuda.synchronize()
        assert int(res[0]) == nbytes and torch.equal(raw, back), (int(res[0]) & 0xFFFFFFFF, int(csize[0]) & 0xFFFFFFFF)
        out = {"case": name, "raw_MB": round(nbytes / 1e6, 2), "ratio": round(nbytes / int(csize[0]), 4)}
        c.profile_reset()
        c.profile(True)
        for _ in range(5):
            enc()
            dec()
        c.profile(False)
        out["launch_groups_ms"] = {k: round(v[1] / max(v[0], 1), 4) for k, v in c.profile_read().items()}
        for label, fn in (("encode", enc), ("decode", dec)):
            k = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(k):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / k
            out[label + "_ms"] = round(ms, 4)
            out[label + "_GBps"] = round(nbytes / ms / 1e6, 2)
        # the single-buffer host API, as a C caller uses it: the caller's buffers exist and have been touched (an HDF5 filter's chunk
        # buffers, a reader's sample array) -- the Python wrappers in vbz.py allocate a fresh array per call and copy the result out of
        # it, which for 40 MB costs several times the call (page faults on 10 000 fresh pages, twice)
        h = np.ascontiguousarray(raw[:nbytes].cpu().numpy())
        o2 = _lib.CompressionOptions(zz, size, level, ver)
        hout = np.zeros(cap + 16, np.uint8)
        hback = np.zeros(nbytes, np.uint8)
        n = L.vbz_compress(h.ctypes.data, nbytes, hout.ctypes.data, cap, ctypes.byref(o2))
        m = L.vbz_decompress(hout.ctypes.data, n, hback.ctypes.data, nbytes, ctypes.byref(o2))
        assert m == nbytes and hback.tobytes() == h.tobytes()
        t0 = time.perf_counter()
        for _ in range(10):
            n = L.vbz_compress(h.ctypes.data, nbytes, hout.ctypes.data, cap, ctypes.byref(o2))
        t1 = time.perf_counter()
        for _ in range(10):
            m = L.vbz_decompress(hout.ctypes.data, n, hback.ctypes.data, nbytes, ctypes.byref(o2))
        t2 = time.perf_counter()
        out["host_api_compress_ms"] = round((t1 - t0) / 10 * 1e3, 3)
        out["host_api_decompress_ms"] = round((t2 - t1) / 10 * 1e3, 3)
        t0 = time.perf_counter()
        for _ in range(3):
            f = vbz.compress_raw(h, o2)
        t1 = time.perf_counter()
        for _ in range(3):
            b = vbz.decompress_raw(f, nbytes, o2)
        t2 = time.perf_counter()
        out["python_wrapper_ms"] = [round((t1 - t0) / 3 * 1e3, 3), round((t2 - t1) / 3 * 1e3, 3)]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
