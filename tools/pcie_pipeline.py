#!/usr/bin/env python3
"""What the codec delivers when the data lives in HOST memory (never bench.py's `value`: that is HBM-resident).

    python tools/pcie_pipeline.py [--reads 2048] [--batches 12]

Encode: pinned host int16 reads -> H2D -> compress -> compact on the device -> D2H of the compressed bytes.
Decode: pinned host chunks -> H2D -> decompress -> D2H of the samples.
Three streams (copy in, codec, copy out) and two sets of device buffers keep a batch in each stage; the rates are
whole-pipeline wall clock over `--batches` batches, in MB/s of raw int16 bytes.  Prints one JSON line."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def measure(codec, reads=2048, batches=12):
    """Runs both pipelines on `codec`'s device; returns the result dict (see the module docstring)."""
    from vbz_compression_amd import batch

    class args:  # noqa: N801
        pass

    args.reads, args.batches = reads, batches
    dev = codec.device
    L = codec.L
    opts = codec.options(True, 2, 1, 1)
    copy_opts = codec.options(False, 0, 0, 0)  # integer_size 0, level 0: the batch call is a per-read byte copy
    n = args.reads
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    with torch.cuda.stream(codec.stream):
        lens = codec.synth_lengths(5, 0, n)
        sizes = lens.to(torch.int64) * 2
        off, total = batch.layout(sizes.cpu(), 64)
        off = off.to(dev)
        raw0 = torch.empty(total, dtype=torch.uint8, device=dev)
        codec.synth_signal(5, 0, raw0, off, lens)
        size32 = sizes.to(torch.int32)
        caps = torch.tensor([L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
        coff, ctotal = batch.layout(caps, 64)
        coff = coff.to(dev)
        cap32 = caps.to(torch.int32).to(dev)
    torch.cuda.synchronize()
    raw_bytes = int(sizes.sum())
    h_raw = torch.empty(total, dtype=torch.uint8).pin_memory()
    h_raw.copy_(raw0)
    h_back = torch.empty(total, dtype=torch.uint8).pin_memory()
    h_comp = torch.empty(raw_bytes, dtype=torch.uint8).pin_memory()  # compressed bytes of a batch, dense

    sets = []
    for _ in range(2):
        sets.append(dict(raw=torch.empty(total, dtype=torch.uint8, device=dev), comp=torch.empty(ctotal, dtype=torch.uint8, device=dev),
                         dense=torch.empty(raw_bytes + 64, dtype=torch.uint8, device=dev), csize=torch.zeros(n, dtype=torch.int32, device=dev),
                         doff=torch.zeros(n, dtype=torch.int64, device=dev), res=torch.zeros(n, dtype=torch.int32, device=dev),
                         tot=torch.zeros(1, dtype=torch.int64, device=dev), h_tot=torch.zeros(1, dtype=torch.int64).pin_memory(),
                         e_in=torch.cuda.Event(), e_codec=torch.cuda.Event(), e_out=torch.cuda.Event()))

    def encode_pipeline(k):
        comp_total = 0
        t0 = time.perf_counter()
        for i in range(k + 2):
            if i < k:  # stage 1: host -> device
                S = sets[i % 2]
                with torch.cuda.stream(s_in):
                    s_in.wait_event(S["e_out"])  # the set's previous results have left
                    S["raw"].copy_(h_raw, non_blocking=True)
                    S["e_in"].record()
            if 1 <= i <= k:  # stage 2: code and compact
                S = sets[(i - 1) % 2]
                with torch.cuda.stream(codec.stream):
                    codec.stream.wait_event(S["e_in"])
                    codec.compress(S["raw"], off, size32, S["comp"], coff, cap32, S["csize"], opts)
                    sz = S["csize"].to(torch.int64)
                    S["doff"].copy_(torch.cumsum(sz, 0) - sz)
                    S["tot"].copy_(sz.sum().reshape(1))
                    codec.compress(S["comp"], coff, S["csize"], S["dense"], S["doff"], S["csize"], S["res"], copy_opts)
                    S["h_tot"].copy_(S["tot"], non_blocking=True)
                    S["e_codec"].record()
            if i >= 2:  # stage 3: device -> host, exactly the bytes produced
                S = sets[(i - 2) % 2]
                S["e_codec"].synchronize()
                tot = int(S["h_tot"][0])
                comp_total += tot
                with torch.cuda.stream(s_out):
                    h_comp[:tot].copy_(S["dense"][:tot], non_blocking=True)
                    S["e_out"].record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, comp_total // k

    def decode_pipeline(k, comp_bytes, h_doff, h_csize):
        d_doff, d_csize = h_doff.to(dev), h_csize.to(dev)
        t0 = time.perf_counter()
        for i in range(k + 2):
            if i < k:
                S = sets[i % 2]
                with torch.cuda.stream(s_in):
                    s_in.wait_event(S["e_out"])
                    S["dense"][:comp_bytes].copy_(h_comp[:comp_bytes], non_blocking=True)
                    S["e_in"].record()
            if 1 <= i <= k:
                S = sets[(i - 1) % 2]
                with torch.cuda.stream(codec.stream):
                    codec.stream.wait_event(S["e_in"])
                    codec.decompress(S["dense"], d_doff, d_csize, S["raw"], off, size32, S["res"], opts)
                    S["e_codec"].record()
            if i >= 2:
                S = sets[(i - 2) % 2]
                with torch.cuda.stream(s_out):
                    s_out.wait_event(S["e_codec"])
                    h_back.copy_(S["raw"], non_blocking=True)
                    S["e_out"].record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    encode_pipeline(3)
    t_enc, comp_bytes = encode_pipeline(args.batches)
    S = sets[(args.batches - 1) % 2]
    h_doff, h_csize = S["doff"].cpu(), S["csize"].cpu()
    assert bool((S["res"] == S["csize"]).all())
    decode_pipeline(3, comp_bytes, h_doff, h_csize)
    t_dec = decode_pipeline(args.batches, comp_bytes, h_doff, h_csize)
    ok = bool((sets[(args.batches - 1) % 2]["res"] == size32).all()) and torch.equal(h_back, h_raw)
    # the copies alone, for scale
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        sets[0]["raw"].copy_(h_raw, non_blocking=True)
    torch.cuda.synchronize()
    h2d = 4 * total / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    for _ in range(4):
        h_back.copy_(sets[0]["raw"], non_blocking=True)
    torch.cuda.synchronize()
    d2h = 4 * total / (time.perf_counter() - t0) / 1e9
    k = args.batches
    return {
        "workload": "%d reads per batch, %d batches, pinned host memory both ends" % (n, k),
        "round_trip_ok": ok,
        "encode_MBps": round(k * raw_bytes / t_enc / 1e6, 1),
        "decode_MBps": round(k * raw_bytes / t_dec / 1e6, 1),
        "encode_decode_MBps": round(k * raw_bytes / (t_enc + t_dec) / 1e6, 1),
        "ratio": round(raw_bytes / comp_bytes, 4),
        "h2d_GBps": round(h2d, 1), "d2h_GBps": round(d2h, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--batches", type=int, default=12)
    args = ap.parse_args()
    from vbz_compression_amd import batch

    torch.cuda.set_device(0)
    out = measure(batch.GpuCodec(0), args.reads, args.batches)
    print(json.dumps(out))
    return 0 if out["round_trip_ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
