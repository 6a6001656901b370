#!/usr/bin/env python3
"""One read per call through the batched entry points, device-resident, for a few read lengths: milliseconds per compress and
per decompress call (HIP events around 50 calls each), the ratio, and the same read through the single-buffer host API.  Run it with VBZ_HIP_SEGMENTED=0 and =1 to compare the
one-wavefront path with the large-read path at sizes below the shape rule's threshold:

    VBZ_HIP_SEGMENTED=1 python tools/time_one_read.py [samples ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbz_compression_amd import batch

codec = batch.GpuCodec(0)
torch.cuda.set_stream(codec.stream)
opts = codec.options(True, 2, 1, 1)
L = codec.L
for n in [int(x) for x in sys.argv[1:]] or [25_000, 50_000, 100_000, 200_000, 400_000]:
    lens = torch.tensor([n], dtype=torch.int32, device="cuda")
    off = torch.zeros(1, dtype=torch.int64, device="cuda")
    raw = torch.empty(2 * n + 64, dtype=torch.uint8, device="cuda")
    codec.synth_signal(5, 3, raw, off, lens)
    size = torch.tensor([2 * n], dtype=torch.int32, device="cuda")
    cap = L.vbz_max_compressed_size(2 * n, ctypes.byref(opts))
    comp = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
    cap_t = torch.tensor([cap], dtype=torch.int32, device="cuda")
    cs = torch.zeros(1, dtype=torch.int32, device="cuda")
    back = torch.empty_like(raw)
    res = torch.zeros(1, dtype=torch.int32, device="cuda")
    src = raw[: 2 * n]

    def enc():
        codec.compress(src, off, size, comp[:cap], off, cap_t, cs, opts)

    def dec():
        codec.decompress(comp[:cap], off, cs, back[: 2 * n], off, size, res, opts)

    for _ in range(3):
        enc()
        dec()
    torch.cuda.synchronize()
    assert torch.equal(raw[: 2 * n], back[: 2 * n])
    out = []
    for fn in (enc, dec):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50)
    # the same read through vbz_compress / vbz_decompress of include/vbz.h: host memory in and out, the caller reuses its buffers
    import time

    import numpy as np

    h = raw[: 2 * n].cpu().numpy()
    cbuf = np.zeros(cap + 16, np.uint8)
    dbuf = np.zeros(2 * n, np.uint8)
    k = 30
    for it in range(k + 3):
        if it == 3:
            t0 = time.perf_counter()
        m = L.vbz_compress(h.ctypes.data, 2 * n, cbuf.ctypes.data, cap, ctypes.byref(opts))
    t1 = time.perf_counter()
    for it in range(k + 3):
        if it == 3:
            t1b = time.perf_counter()
        q = L.vbz_decompress(cbuf.ctypes.data, m, dbuf.ctypes.data, 2 * n, ctypes.byref(opts))
    t2 = time.perf_counter()
    assert q == 2 * n and dbuf.tobytes() == h.tobytes()
    print("%7d samples: compress %.3f ms, decompress %.3f ms, ratio %.4f; vbz_compress %.3f ms, vbz_decompress %.3f ms (VBZ_HIP_SEGMENTED=%s)"
          % (n, out[0], out[1], 2 * n / int(cs[0]), (t1 - t0) / k * 1e3, (t2 - t1b) / k * 1e3, os.environ.get("VBZ_HIP_SEGMENTED", "unset")))
