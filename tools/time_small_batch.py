#!/usr/bin/env python3
"""A small batch of equal reads through the batched entry points, device-resident: milliseconds per compress and per decompress call
(HIP events around 50 calls each).  Run it with VBZ_HIP_SEGMENTED=0 and =1 to compare the one-wavefront path with the large-read path
below the shape rule's thresholds (use_segments in vbz_api.hip):

    VBZ_HIP_SEGMENTED=1 python tools/time_small_batch.py <reads> <samples per read> [...]"""
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
args = [int(x) for x in sys.argv[1:]] or [64, 10000]
for n, samples in zip(args[0::2], args[1::2]):
    lens = torch.full((n,), samples, dtype=torch.int32, device="cuda")
    sizes = lens.to(torch.int64) * 2
    off, total = batch.layout(sizes.cpu(), 64)
    off = off.cuda()
    raw = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
    codec.synth_signal(5, 7, raw, off, lens)
    s32 = sizes.to(torch.int32)
    cap = L.vbz_max_compressed_size(2 * samples, ctypes.byref(opts))
    coff, ctotal = batch.layout([cap] * n, 64)
    coff = coff.cuda()
    cap32 = torch.full((n,), cap, dtype=torch.int32, device="cuda")
    comp = torch.empty(ctotal + 64, dtype=torch.uint8, device="cuda")
    cs = torch.zeros(n, dtype=torch.int32, device="cuda")
    back = torch.empty_like(raw)
    res = torch.zeros(n, dtype=torch.int32, device="cuda")

    def enc():
        codec.compress(raw[:total], off, s32, comp[:ctotal], coff, cap32, cs, opts)

    def dec():
        codec.decompress(comp[:ctotal], coff, cs, back[:total], off, s32, res, opts)

    for _ in range(3):
        enc()
        dec()
    torch.cuda.synchronize()
    assert torch.equal(raw[:total], back[:total])
    out = []
    for fn in (enc, dec):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50)
    print("%5d reads of %7d samples: compress %.3f ms, decompress %.3f ms, ratio %.4f (VBZ_HIP_SEGMENTED=%s)"
          % (n, samples, out[0], out[1], float(sizes.sum()) / float(cs.sum()), os.environ.get("VBZ_HIP_SEGMENTED", "unset")))
