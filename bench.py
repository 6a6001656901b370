#!/usr/bin/env python3
"""bench.py -- headline benchmark of the VBZ int16 hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- vbz_compress then vbz_decompress semantics for every read of
one batch (delta zig-zag + streamvbyte + zstd-format entropy stage, and back) -- over one batch of
synthetic int16 reads already resident in HBM.  Workload = BASELINE.json configs[1]: synthetic
int16 reads of ~100k samples (SURVEY.md 8d generator, seed 5), `--reads` reads per batch.
Prints ONE JSON line (metric: MB/s of raw int16 bytes through encode+decode).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# reads per batch.  A launch ends with a tail of partly idle CUs (about one frame's latency per kernel), so batches are
# large: 65536 reads = 13 GB of raw signal, ~140 GB of HBM for two resident batches with their worst-case output
# slots and the library's scratch.  Measured on one box: 8192 reads 365 GB/s, 16384: 385, 32768: 395, 65536: 411.
DEFAULT_READS = 65536
PEAK_HBM_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy reaches


def cpu_baseline(min_seconds=8.0, n_reads=2048):
    """The oracle (port of the reference CPU path + the pinned libzstd, dlopen'd) timed on this box's
    host cores on a bounded sample of the same workload: reads [0, n_reads) of the same generator,
    encode+decode, one pthread per hardware thread (reads are independent; the reference has no
    internal threading).  The oracle is checker code: here it is only the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = os.cpu_count() or 1
    opts = O.options(True, 2, 1, 1)
    one = O.bench_roundtrip(min(n_reads, 64), 1, 1.0, opts)
    allc = O.bench_roundtrip(n_reads, cores, min_seconds, opts)
    out = {
        "value": round(allc["raw_bytes"] / allc["best_s"] / 1e6, 1),
        "unit": "MB/s",
        "cores": cores,
        "kind": "port",
        "sample": "reads 0..%d of the same generator (%.1f MB raw), encode+decode, %d threads (all hardware threads of the host), "
                  "libzstd %s level 1, best of %d passes; one thread: %.1f MB/s"
        % (n_reads - 1, allc["raw_bytes"] / 1e6, cores, (O.lib().vbo_zstd_version() or b"?").decode(), allc["passes"],
           one["raw_bytes"] / one["best_s"] / 1e6),
        "ratio": round(allc["raw_bytes"] / allc["comp_bytes"], 4),
        "encode_share": round(allc["enc_thread_s"] / (allc["enc_thread_s"] + allc["dec_thread_s"]), 3),
    }
    # BASELINE.json's target is stated against a single socket: time one socket's hardware threads as well
    # (the worker pthreads inherit the affinity set here)
    try:
        socket0 = []
        for cpu in sorted(os.sched_getaffinity(0)):
            with open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % cpu) as f:
                if int(f.read()) == 0:
                    socket0.append(cpu)
        if socket0 and len(socket0) < cores:
            saved = os.sched_getaffinity(0)
            os.sched_setaffinity(0, socket0)
            try:
                s0 = O.bench_roundtrip(n_reads, len(socket0), min_seconds / 2, opts)
            finally:
                os.sched_setaffinity(0, saved)
            out["single_socket"] = {"value": round(s0["raw_bytes"] / s0["best_s"] / 1e6, 1), "unit": "MB/s", "cores": len(socket0)}
    except OSError:
        pass
    return out


def committed_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest rocprofv3 PMC summary committed under profiles/
    (tools/summarize_profile.py: separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 read correction).
    bench.py cannot run the profiler on itself.  Returns (bytes, file, reads per launch of the profiled run): traffic is
    proportional to the number of reads, so it is scaled to this run's batch when the two differ."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.csv")))
    for path in reversed(files):
        for r in csv.DictReader(open(path)):
            if r["kernel"].startswith(kernel):
                # reads per launch of the profiled run: from the bench line committed with the same tag
                reads = 8192
                try:
                    with open(path.replace("_hbm_traffic.csv", "_bench.json")) as f:
                        reads = int(json.load(f)["config"]["reads_per_step"])
                except (OSError, KeyError, ValueError):
                    pass
                return int(float(r["hbm_MB_per_launch"]) * 1e6), os.path.basename(path), reads
    return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=DEFAULT_READS, help="reads per batch (one batch per step)")
    ap.add_argument("--resident", type=int, default=2, help="distinct batches kept in HBM and cycled")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the codec has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=dev)
    from vbz_compression_amd import batch, shard

    codec = batch.GpuCodec(local_rank)
    torch.cuda.set_stream(codec.stream)  # everything below (generation, events, kernels) runs on the codec's stream
    opts = codec.options(True, 2, 1, 1)
    L = codec.L
    n = args.reads
    if n == DEFAULT_READS:  # the default needs ~140 GB: step down on a GPU that does not have it free
        free = torch.cuda.mem_get_info(dev)[0]
        while n > 8192 and free < n * 2.4e6:
            n //= 2

    # ---- resident batches: rank r owns batches r, r+W, ... of the global read table (weak scaling)
    batches = []
    for b in range(args.resident):
        gb = rank + b * world  # global batch index
        first = gb * n
        lens = codec.synth_lengths(5, first, n)
        sizes = lens.to(torch.int64) * 2
        off, total = batch.layout(sizes.cpu(), 64)
        raw = torch.empty(total, dtype=torch.uint8, device=dev)
        off = off.to(dev)
        codec.synth_signal(5, first, raw, off, lens)
        size32 = sizes.to(torch.int32)
        caps = torch.tensor([L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
        coff, ctotal = batch.layout(caps, 64)
        comp = torch.empty(ctotal, dtype=torch.uint8, device=dev)
        batches.append(
            dict(raw=raw, off=off, size=size32, comp=comp, coff=coff.to(dev), cap=caps.to(torch.int32).to(dev),
                 csize=torch.zeros(n, dtype=torch.int32, device=dev), back=torch.empty_like(raw),
                 res=torch.zeros(n, dtype=torch.int32, device=dev), raw_bytes=int(sizes.sum()), samples=int(lens.sum()))
        )
    torch.cuda.synchronize()

    def step(i, timed_parts=None):
        B = batches[i % len(batches)]
        codec.compress(B["raw"], B["off"], B["size"], B["comp"], B["coff"], B["cap"], B["csize"], opts)
        if timed_parts is not None:
            timed_parts[0].record()
        codec.decompress(B["comp"], B["coff"], B["csize"], B["back"], B["off"], B["size"], B["res"], opts)
        return B

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # correctness of what is being timed: every read round-trips, on the device
    B = batches[0]
    if not os.environ.get("VBZ_BENCH_KERNEL_EXPERIMENT"):  # set only to time deliberately broken kernel variants
        assert bool((B["res"] == B["size"]).all()), "decode failed for some read"
        assert torch.equal(B["raw"], B["back"]), "round trip mismatch"

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    codec.profile_reset()
    codec.profile(True)
    ev0 = torch.cuda.Event(enable_timing=True)
    evm = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    eve = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    raw_bytes = 0
    samples = 0
    for i in range(args.steps):
        if i == 0:
            ev0.record()
        Bi = step(i, (evm[i],))
        eve[i].record()
        raw_bytes += Bi["raw_bytes"]
        samples += Bi["samples"]
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    codec.profile(False)
    elapsed = shard.max_over_ranks(elapsed, dev)
    prof = codec.profile_read()
    comp_bytes = int(batches[0]["csize"].to(torch.int64).sum())
    ratio = batches[0]["raw_bytes"] / comp_bytes
    table, _ = shard.exchange_tallies(args.steps * n, raw_bytes, comp_bytes * args.steps, dev)
    total_raw = int(table[:, 1].sum())

    # encode / decode split from the events on torch's stream (the codec launches on it)
    enc_ms = dec_ms = 0.0
    prev = ev0
    for i in range(args.steps):
        enc_ms += prev.elapsed_time(evm[i])
        dec_ms += evm[i].elapsed_time(eve[i])
        prev = eve[i]

    if rank == 0:
        # ---- roofline of the dominant kernel (largest total time over the timed region)
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        name, (launches, tot_ms) = dom
        svb_bytes = None
        per_sample = {
            # algorithmic bytes per int16 sample handled by ONE launch of that kernel (DESIGN.md "Kernels")
            "svb_encode": 2.0 + 1.261,                             # read raw, write svb stream
            "svb_decode": 1.261 + 2.0,
            "zstd_encode": 1.261 + 2.0 / ratio,                   # read svb stream, write frame
            "zstd_decode": 2.0 / ratio + 1.261,
        }.get(name, 2.0)
        traffic, traffic_src, traffic_reads = committed_traffic(name + "_kernel")
        if traffic is not None and traffic_reads != n:
            traffic = int(traffic * (n / traffic_reads))
            traffic_src = "%s, scaled from %d to %d reads per launch" % (traffic_src, traffic_reads, n)
        avg_ms = tot_ms / max(launches, 1)
        alg_bytes = per_sample * (samples / args.steps)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": "MB/s encode+decode, int16 signal, 1/2/4/8 MI355X vs CPU; ratio preserved",
            "value": round(total_raw / elapsed / 1e6, 1),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: synthetic int16 reads of ~100k samples (SURVEY 8d generator, seed 5), "
                            "%d reads (%.2f GB raw) per step per GPU, zig-zag + svb + zstd-format stage, encode then decode, inputs resident in HBM"
                            % (n, batches[0]["raw_bytes"] / 1e9),
                "reads_per_step": n,
                "options": "zigzag=1,integer_size=2,zstd_level=1,vbz_version=1",
                "parallelism": "reads sharded across %d GPU(s), no data-path collective" % world,
            },
            "ratio": round(ratio, 4),
            "encode_MBps": round(raw_bytes / (enc_ms * 1e-3) / 1e6, 1),
            "decode_MBps": round(raw_bytes / (dec_ms * 1e-3) / 1e6, 1),
            "kernels_ms_per_launch": {k: round(v[1] / max(v[0], 1), 4) for k, v in prof.items()},
            "roofline": {
                "bound": "hbm",
                "kernel": name,
                "achieved": round(achieved, 2),
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": round(achieved / PEAK_HBM_GBS, 5),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "avg_launch_ms": round(avg_ms, 4),
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
            if "single_socket" in out["cpu_baseline"]:
                out["vs_single_socket_cpu"] = round(out["value"] / out["cpu_baseline"]["single_socket"]["value"], 2)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
