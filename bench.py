#!/usr/bin/env python3
"""bench.py -- headline benchmark of the VBZ int16 hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- vbz_compress then vbz_decompress semantics for every read of
one batch (delta zig-zag + streamvbyte + zstd-format entropy stage, and back) -- over one batch of
synthetic int16 reads already resident in HBM.  Workload = BASELINE.json configs[1]: synthetic
int16 reads of ~100k samples (SURVEY.md 8d generator, seed 5), `--reads` reads per batch; as many DISTINCT
batches as HBM holds are kept resident (about one million reads on a 288 GB MI355X), every one of them is
round-trip verified before the timed region, and the timed steps cycle over them.
Prints ONE JSON line (metric: MB/s of raw int16 bytes through encode+decode).

Multi-GPU: `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a launcher: it starts N
rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set) BEFORE anything touches the GPU, waits
for them and relays rank 0's JSON line.  Under torchrun (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# reads per batch.  A launch ends with a tail of partly idle CUs (about one frame's latency per kernel), so batches are
# large: 65536 reads = 13 GB of raw signal.  Measured on one box: 8192 reads 365 GB/s, 16384: 385, 32768: 395, 65536: 411.
DEFAULT_READS = 65536
PEAK_HBM_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy reaches
METRIC = "MB/s encode+decode, int16 signal, 1/2/4/8 MI355X vs CPU; ratio preserved"
SVB_BYTES_PER_SAMPLE = 1.261  # svb stream bytes per int16 sample of this workload


# ---------------------------------------------------------------------------------------------------------------------
# launcher (no GPU call may happen before or inside it)
# ---------------------------------------------------------------------------------------------------------------------
def kfd_gpu_count():
    """GPUs the kernel driver knows (topology nodes with SIMDs; CPUs are nodes without), or None if sysfs does not say."""
    import glob

    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    count = 0
    for path in nodes:
        try:
            with open(path) as f:
                for ln in f:
                    k, _, v = ln.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        count += 1
        except (OSError, ValueError):
            return None
    return count


def launch_ranks(args, argv):
    """Start one rank process per GPU, wait for all of them, relay rank 0's JSON line.  Every rank's stderr goes to a file of its own
    (gpurun_out/ranks/rank<k>.err when that directory can be made, else a temporary one): a rank that fails takes its peers down
    with timeouts, and it is the FIRST failing rank's last lines that say why -- they are relayed on the launcher's stderr."""
    import tempfile

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    logdir = os.path.join(ROOT, "gpurun_out", "ranks")
    try:
        os.makedirs(logdir, exist_ok=True)
    except OSError:
        logdir = tempfile.mkdtemp(prefix="vbz_bench_ranks_")
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = open(os.path.join(logdir, "rank%d.err" % r), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=log))
    out0 = procs[0].communicate()[0].decode()
    # the order the ranks end in: the first one to fail is the cause, the others follow it down
    first_bad, rcs = None, [None] * len(procs)
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rc = p.poll()
                if rc is not None:
                    rcs[r] = rc
                    if rc != 0 and first_bad is None:
                        first_bad = r
        time.sleep(0.05)
    line = ""
    for ln in out0.splitlines():
        if ln.startswith("{"):
            line = ln
    if line:
        print(line, flush=True)
    tails = []
    for r, log in enumerate(logs):
        log.flush()
        log.seek(0)
        tails.append(log.read()[-2000:])
        log.close()
    if first_bad is not None or not line:
        sys.stderr.write("bench.py launcher: rank exit codes %s%s; per-rank stderr in %s\n" % (rcs, "" if line else "; rank 0 printed no JSON line", logdir))
        who = first_bad if first_bad is not None else 0
        sys.stderr.write("---- rank %d (the first to fail) ----\n%s\n" % (who, tails[who]))
        return 1
    if tails[0].strip():   # (warnings of a run that went through: rank 0's, as before)
        sys.stderr.write(tails[0])
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline leg
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline_u32(n_buffers, count, min_seconds=6.0):
    """CPU baseline of the config-4 workload: the oracle (streamvbyte restated from its published format + the pinned
    libzstd at level 3) over `n_buffers` buffers of `count` uint32 values, one persistent thread per usable CPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores, quota_note = usable_cpus()
    opts = O.options(False, 4, 3, 0)
    threads = min(cores, n_buffers)
    one = O.bench_roundtrip(1, 1, 1.0, opts, u32_count=count)
    allc = O.bench_roundtrip(n_buffers, threads, min_seconds, opts, u32_count=count)
    return {
        "value": round(allc["raw_bytes"] / allc["best_s"] / 1e6, 1), "unit": "MB/s", "cores": threads, "kind": "port",
        "sample": "%d buffers of %d uint32 values of the same generator (%.1f MB raw), encode+decode, %d persistent threads (%s; one buffer "
                  "per thread at a time: the reference has no internal threading), scalar svb + libzstd %s level 3, best of %d timed passes; "
                  "one thread: %.1f MB/s" % (n_buffers, count, allc["raw_bytes"] / 1e6, threads, quota_note,
                                             (O.lib().vbo_zstd_version() or b"?").decode(), allc["passes"], one["raw_bytes"] / one["best_s"] / 1e6),
        "one_thread": round(one["raw_bytes"] / one["best_s"] / 1e6, 1),
        "ratio": round(allc["raw_bytes"] / allc["comp_bytes"], 4),
    }


def decode_reference_frames(codec, n_reads=16384, launches=5):
    """Decode rate on frames THE REFERENCE wrote: every vbz file in existence was written by
    vbz_compress -> ZSTD_compress (vbz/vbz.cpp:194-207: one 128 KB block per read, ~1 100 general sequences, four long Huffman
    streams), and such frames carry none of this library's decoder hints.  Frames of reads [0, n_reads) of the same generator are
    written by the oracle (the reference path restated + the pinned libzstd, level 1) on the host's cores, decoded by the batched
    entry point `launches` times, and every decoded read is compared with the generator's samples on the device.  Never `value`."""
    import concurrent.futures

    import numpy as np
    import torch

    from vbz_compression_amd import batch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    dev = codec.device
    oo = O.options(True, 2, 1, 1)
    opts = codec.options(True, 2, 1, 1)
    O.lib()

    def make(i):
        a = O.synth_signal(5, i, O.synth_read_length(5, i))
        return a.nbytes, O.compress(a, oo)

    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(max(1, usable_cpus()[0])) as ex:   # (ctypes calls release the GIL)
        made = list(ex.map(make, range(n_reads)))
    t_make = time.perf_counter() - t0
    sizes = torch.tensor([m[0] for m in made], dtype=torch.int64)
    fsizes = torch.tensor([len(m[1]) for m in made], dtype=torch.int64)
    foff, ftotal = batch.layout(fsizes, 64)
    arena = np.zeros(ftotal + 64, np.uint8)
    for (_, f), o in zip(made, foff.tolist()):
        arena[o : o + len(f)] = f
    del made
    src = torch.from_numpy(arena).to(dev)
    off, total = batch.layout(sizes, 64)
    want = torch.zeros(total, dtype=torch.uint8, device=dev)
    lens = (sizes // 2).to(torch.int32).to(dev)
    offd = off.to(dev)
    codec.synth_signal(5, 0, want, offd, lens)
    back = torch.zeros(total, dtype=torch.uint8, device=dev)
    res = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    foffd, fs32, s32 = foff.to(dev), fsizes.to(torch.int32).to(dev), sizes.to(torch.int32).to(dev)

    def go():
        codec.decompress(src, foffd, fs32, back, offd, s32, res, opts)

    go()
    torch.cuda.synchronize()
    ok = bool((res == s32).all()) and torch.equal(back, want)
    go()   # (a second untimed call: the first finds out whose frames these are -- and runs as two halves --, this one sizes the context's
    torch.cuda.synchronize()   # buffers for the whole call; the timed calls are the steady state, like the steps behind the headline's warm-up)
    ok = ok and bool((res == s32).all()) and torch.equal(back, want)
    codec.profile_reset()
    codec.profile(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        go()
    e1.record()
    torch.cuda.synchronize()
    codec.profile(False)
    prof = codec.profile_read()
    paths = codec.decode_paths()   # (frames, decoded by the batched own-frame decoder, sequence chains walked ahead) of the last call
    ms = e0.elapsed_time(e1) / launches
    raw = int(sizes.sum())
    per = {k: round(v[1] / max(v[0], 1), 4) for k, v in prof.items()}
    c = float(fsizes.sum()) / (raw / 2)   # frame bytes per sample
    return {
        "MBps": round(raw / (ms * 1e-3) / 1e6, 1), "unit": "MB/s of decoded int16 signal", "reads": n_reads, "raw_MB": round(raw / 1e6, 1), "ms_per_launch": round(ms, 3),
        "kernels_ms_per_launch": per,
        "entropy_stage_ms_per_2048_reads": round(per.get("zstd_decode", 0.0) * 2048 / n_reads, 4),
        "hbm_frac": round((2 + c) * (raw / 2) / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
        "ratio": round(raw / float(fsizes.sum()), 4), "verified": ok,
        "chains_walked_ahead": paths[2],   # frames whose sequence chains zstd_decode_ref.hip walked (one lane per frame)
        "frames": "written by the oracle: the reference path restated + libzstd %s level 1 (one 128 KB block per read, general sequences, "
                  "no decoder hints), %d host threads, %.1f s" % ((O.lib().vbo_zstd_version() or b"?").decode(), usable_cpus()[0], t_make),
        "note": "decode only, inputs resident in HBM; never `value`",
    }


def cpu_baseline(min_seconds=8.0, n_reads=16384):
    """The oracle (port of the reference CPU path: the int16 zig-zag stage in SSSE3 form like the reference's hot path --
    oracle/vbz_oracle_simd.c, own code, byte-identical to the scalar restatement -- plus the pinned libzstd, dlopen'd) timed on
    this box's host cores on a bounded sample of the same workload: reads [0, n_reads) of the same generator, encode+decode,
    persistent pthreads claiming reads from a queue (reads are independent; the reference has no internal threading),
    verification in an untimed pass.  The oracle is checker code: here it is only the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    hw = os.cpu_count() or 1
    cores, quota_note = usable_cpus()
    if cores < 32:  # a small host (or a small CPU quota): keep the leg at ~10-30 s of CPU work
        n_reads = min(n_reads, 256 * cores)
    opts = O.options(True, 2, 1, 1)
    simd = O.simd_available()
    one_scalar = O.bench_roundtrip(min(n_reads, 64), 1, 1.0, opts)
    one = O.bench_roundtrip(min(n_reads, 64), 1, 1.0, opts, simd=simd)
    allc = O.bench_roundtrip(n_reads, cores, min_seconds, opts, simd=simd)
    one_MBps = one["raw_bytes"] / one["best_s"] / 1e6
    out = {
        "value": round(allc["raw_bytes"] / allc["best_s"] / 1e6, 1),
        "unit": "MB/s",
        "cores": cores,
        "kind": "port",
        "sample": "reads 0..%d of the same generator (%.1f MB raw), encode+decode, %d persistent threads (%s) claiming reads from a "
                  "queue, %s svb + libzstd %s level 1, verification in an untimed pass, best of %d timed passes; one thread: %.1f MB/s "
                  "(%.1f with the scalar svb)"
        % (n_reads - 1, allc["raw_bytes"] / 1e6, cores, quota_note, "SSSE3 (own code, the reference's class of path)" if simd else "scalar",
           (O.lib().vbo_zstd_version() or b"?").decode(), allc["passes"], one_MBps, one_scalar["raw_bytes"] / one_scalar["best_s"] / 1e6),
        "one_thread": round(one_MBps, 1),
        "one_thread_scalar_svb": round(one_scalar["raw_bytes"] / one_scalar["best_s"] / 1e6, 1),
        "host_hw_threads": hw,
        "ratio": round(allc["raw_bytes"] / allc["comp_bytes"], 4),
        "encode_share": round(allc["enc_thread_s"] / (allc["enc_thread_s"] + allc["dec_thread_s"]), 3),
    }
    # BASELINE.json's target is stated against a single socket.  When the process cannot use one (a CPU quota, an affinity
    # mask) it is EXTRAPOLATED: one thread's rate x the physical cores of socket 0 (an upper bound: perfect scaling)
    try:
        cores0 = set()
        import glob as _glob

        for path in _glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/physical_package_id"):
            with open(path) as f:
                if int(f.read()) != 0:
                    continue
            with open(os.path.join(os.path.dirname(path), "core_id")) as f:
                cores0.add(int(f.read()))
        if cores0 and cores < len(cores0):
            out["single_socket"] = {"value": round(one_MBps * len(cores0), 1), "unit": "MB/s", "cores": len(cores0),
                                    "how": "extrapolated = one_thread x physical cores of socket 0 (the process may use %d CPUs' worth of time)" % cores}
    except (OSError, ValueError):
        pass
    # BASELINE.json's target is stated against a single socket: time one socket's hardware threads as well
    # (the worker pthreads inherit the affinity set here)
    try:
        socket0 = []
        for cpu in sorted(os.sched_getaffinity(0)):
            with open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % cpu) as f:
                if int(f.read()) == 0:
                    socket0.append(cpu)
        if socket0 and len(socket0) < cores and cores == hw:  # only meaningful when the process may use the whole host
            saved = os.sched_getaffinity(0)
            os.sched_setaffinity(0, socket0)
            try:
                s0 = O.bench_roundtrip(n_reads, len(socket0), min_seconds / 2, opts, simd=simd)
            finally:
                os.sched_setaffinity(0, saved)
            out["single_socket"] = {"value": round(s0["raw_bytes"] / s0["best_s"] / 1e6, 1), "unit": "MB/s", "cores": len(socket0), "how": "measured"}
    except OSError:
        pass
    return out


def usable_cpus():
    """Hardware threads this process can really keep busy: the affinity mask, capped by the cgroup CPU quota (a container
    may see 256 CPUs and be allowed 8 CPUs' worth of time -- short bursts then look 10x faster than sustained work)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = "all %d hardware threads in the affinity mask" % n
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        k = max(1, int(quota + 0.5))
        note = "cgroup CPU quota %.1f of %d hardware threads" % (quota, n)
        n = k
    return n, note


# The library's timed labels (vbz_api.hip `Timed`) and the kernels behind each, as rocprofv3 names them (prefix match).  Every kernel of
# libvbz_hip.so must appear here: `profile_tables` refuses a committed profile in which a library kernel holding more than 1 % of the
# library's time belongs to no label (round 5's line dropped zstd_plan_kernel that way and under-reported the encode traffic).
LABEL_KERNELS = {
    "svb_encode": ("svb_encode_kernel", "svb_seg_encode"),
    "zstd_encode": ("zstd_plan_kernel", "zstd_pack_kernel", "zstd_encode_kernel", "zstd_span_", "period_probe_kernel"),
    "zstd_decode": ("fast_scan_kernel", "fast_weights_kernel", "fast_streams_kernel", "fast_runs_kernel", "zstd_decode_kernel", "ref_chain_kernel",
                    "ref_lit_scan_kernel", "ref_pieces_kernel", "zstd_dspan_"),
    "svb_decode": ("svb_decode_kernel", "svb_seg_decode"),
    "plan_scratch": ("plan_scratch_kernel",),
    "route": ("route_", "seg_plan_kernel", "validate_batch_kernel", "parse_sized_kernel", "hand_back_kernel", "copy_bytes_kernel"),
}
# kernels in a profile of `python bench.py` that are NOT the library's: the generator, torch's fills / compares / reductions, runtime copies
NOT_LIBRARY = ("at::native::", "synth_", "__amd_rocclr", "void at::", "Cijk_", "rccl", "nccl")
ENCODE_LABELS, DECODE_LABELS = ("svb_encode", "zstd_encode"), ("zstd_decode", "svb_decode")


def kernel_label(name):
    for label, prefixes in LABEL_KERNELS.items():
        if any(name.startswith(p) for p in prefixes):
            return label
    return None


def profile_tables():
    """Per-kernel launch time and HBM traffic of the newest rocprofv3 summary pair committed under profiles/ (tools/summarize_profile.py:
    `--kernel-trace --stats`, and separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes with the gfx950 read correction): bench.py cannot run
    the profiler on itself.  Returns ({kernel: {"label", "ms", "hbm_bytes"}}, file tag, reads per launch of the profiled run) for the main
    launches (per-read routing's small-grid second group is listed apart by the summariser and left out).  Raises if a library kernel with
    more than 1 % of the library's time has no label."""
    import csv
    import glob

    for tpath in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.csv")))):
        if "config" in os.path.basename(tpath):
            continue
        spath = tpath.replace("_hbm_traffic.csv", "_kernel_stats.csv")
        if not os.path.exists(spath):
            continue
        rows = {}
        for r in csv.DictReader(open(spath)):
            k = r["kernel"]
            if "second launch group" in k or any(k.startswith(p) for p in NOT_LIBRARY):
                continue
            rows[k] = {"label": kernel_label(k), "ms": float(r["avg_ms"]), "total_ms": float(r["total_ms"]), "hbm_bytes": None}
        for r in csv.DictReader(open(tpath)):
            k = r["kernel"]
            if k in rows:
                rows[k]["hbm_bytes"] = int(float(r["hbm_MB_per_launch"]) * 1e6)
        lib_total = sum(v["total_ms"] for v in rows.values()) or 1.0
        orphans = [k for k, v in rows.items() if v["label"] is None and v["total_ms"] > 0.01 * lib_total]
        if orphans:
            raise SystemExit("bench.py: %s holds kernels with more than 1 %% of the library's time that LABEL_KERNELS attributes to no label: %s"
                             % (os.path.basename(spath), orphans))
        reads = 8192
        try:
            with open(tpath.replace("_hbm_traffic.csv", "_bench.json")) as f:
                reads = int(json.load(f)["config"]["reads_per_step"])
        except (OSError, KeyError, ValueError):
            pass
        return rows, os.path.basename(tpath).replace("_hbm_traffic.csv", ""), reads
    return {}, None, None


def committed_traffic_large(direction, nbuf):
    """HBM bytes per step of one direction of `--workload config4` from profiles/*_config4_hbm_traffic.csv (the profiled run
    had `buffers_per_step` buffers: traffic is proportional to the bytes coded, so it is scaled to this run's).  A kernel
    belongs to the encode direction if its name says encode or span_ (span_plan / span_finish / span_compact), to the decode
    direction if it says decode or dspan; launches per step = its launches / the launches of the once-per-step plan kernel."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_config4_hbm_traffic.csv")))
    if not files:
        return None, None
    path = files[-1]
    rows = list(csv.DictReader(open(path)))
    enc = lambda k: "encode" in k or "zstd_span_" in k   # noqa: E731
    dec = lambda k: "decode" in k or "dspan" in k        # noqa: E731
    steps = 0
    for r in rows:
        if r["kernel"].startswith("zstd_span_plan_kernel" if direction == "encode" else "zstd_dspan_plan_kernel"):
            steps = int(r["launches"])
    if not steps:
        return None, None
    total = sum(float(r["hbm_MB_per_launch"]) * int(r["launches"]) for r in rows if (enc if direction == "encode" else dec)(r["kernel"]))
    prof_buffers = 8
    try:
        with open(path.replace("_hbm_traffic.csv", "_bench.json")) as f:
            prof_buffers = int(json.load(f)["config"]["buffers_per_step"])
    except (OSError, KeyError, ValueError):
        pass
    return int(total * 1e6 / steps * nbuf / prof_buffers), os.path.basename(path)


def max_compressed_sizes(sizes, level):
    """vbz_max_compressed_size (include/vbz.h; reference vbz/vbz.cpp:79-114) for int16 reads, vectorised over a tensor
    of byte sizes: svb bound (n+3)/4 + 4n, ZSTD_COMPRESSBOUND when a level is set, +4."""
    n = sizes // 2
    svb = (n + 3) // 4 + 4 * n
    if level:
        svb = svb + (svb >> 8) + ((svb < (128 << 10)) * (((128 << 10) - svb) >> 11))
    return svb + 4


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist

    from vbz_compression_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with `python bench.py --gpus N` or torchrun --nproc-per-node N)"
                         % (args.gpus, world))
    if args.dry_run:  # launcher / work-queue plumbing only (CPU test of the N > 1 path): no codec, no GPU
        dev = None
        backend = "gloo"
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the codec has no CPU path")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        backend = args.backend
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        assert dist.get_world_size() == args.gpus
    # every rank sits on its own device: the ordinals as gathered over the process group (they go into the line's `collective` object)
    devices = shard.gather_row([local_rank if dev is not None else rank], dev if (world > 1 and backend == "nccl") else None)[:, 0].tolist()
    assert len(devices) == world and len(set(devices)) == world, "two ranks share a device: %s" % devices

    def barrier():
        if world > 1:
            dist.barrier()

    coll_dev = dev if backend == "nccl" else None
    n = args.reads
    opts_tuple = (True, 2, 1, 1)

    if args.dry_run:
        # the same work queue as the real run, on the generator's length formula restated in torch
        fixed_job = args.workload == "config5"
        R = args.resident or 2
        total_reads = args.total_reads if fixed_job else world * R * n
        g = torch.Generator().manual_seed(5)
        lengths = 90000 + torch.randint(0, 20001, (total_reads,), generator=g)
        lengths = shard.share_read_table(lengths if rank == 0 else torch.zeros(total_reads, dtype=torch.int64))
        a, b = shard.partition_reads(lengths, world)[rank]
        if fixed_job:  # the rank's share of the fixed job, in batches of at most n reads, each coded once
            R = max(1, -(-(b - a) // n))
            args.steps = R
        batches = shard.cut_batches(a, b, lengths, R)
        barrier()
        t0 = time.perf_counter()
        raw = sum(int(lengths[x:y].sum()) * 2 for x, y in batches) * args.steps // R
        elapsed = shard.max_over_ranks(time.perf_counter() - t0 + 1e-6)
        table, off = shard.exchange_tallies(sum(y - x for x, y in batches), raw, raw // 2)
        mine_us = int((time.perf_counter() - t0 + 1e-6) * 1e6)
        rows = shard.gather_row([sum(y - x for x, y in batches), raw, mine_us])
        assert int(rows[:, 0].sum()) == total_reads, "the ranks' shares do not add up to the job: %s of %d reads" % (rows[:, 0].tolist(), total_reads)
        coll = dict(shard.describe(backend, devices), reads_per_rank=rows[:, 0].tolist(),
                    per_rank_MBps=[round(float(b_) / max(float(u_), 1.0), 1) for _, b_, u_ in rows.tolist()])
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "collective": coll,
                              "ms_per_step": round(elapsed * 1e3, 3), "higher_is_better": True, "scaling": "strong" if fixed_job else "weak",
                              "vs_baseline": None, "dtype": "int16", "data": "dry-run (no codec: launcher and work-queue plumbing only)",
                              "config": {"workload": "dry-run" + (" of configs[4]: %d reads over %d rank(s)" % (total_reads, world) if fixed_job else ""),
                                         "reads_per_step": n},
                              "rank_imbalance": round(float(table[:, 1].max()) * world / max(float(table[:, 1].sum()), 1.0), 4),
                              "tallies": table.tolist(), "ranges": [list(p) for p in shard.partition_reads(lengths, world)]}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    from vbz_compression_amd import batch

    codec = batch.GpuCodec(local_rank)
    torch.cuda.set_stream(codec.stream)  # everything below (generation, events, kernels) runs on the codec's stream
    if args.workload in ("config4", "config1"):
        run_large(args, codec, dev, rank, world, coll_dev, barrier, shard.describe(backend, devices))
        if world > 1:
            barrier()
            dist.destroy_process_group()
        return 0
    opts = codec.options(*opts_tuple)
    L = codec.L
    host_resident = None
    if world == 1 and not args.no_pcie:
        # host-resident data (never `value`): pinned host -> H2D -> codec -> D2H, three-stage pipeline (tools/pcie_pipeline.py).  Measured
        # FIRST, while this process has one context and three streams: behind the timed region -- the routing context, the side stream, 200 GB
        # of HBM just released -- the same code held the decode leg at half the link (28.7 GB/s of 57; here 50), and so did the tool as a
        # child process while this one kept its queues (round 4's VERDICT, item 9).
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pcie_pipeline

        h = pcie_pipeline.measure(codec, 2048, 16)
        host_resident = {"encode_decode_MBps": h["encode_decode_MBps"], "encode_MBps": h["encode_MBps"], "decode_MBps": h["decode_MBps"],
                         "h2d_GBps": h["h2d_GBps"], "d2h_GBps": h["d2h_GBps"], "round_trip_ok": h["round_trip_ok"],
                         "note": "PCIe-inclusive rate with pinned host buffers both ends (tools/pcie_pipeline.py), measured before the resident batches are built; never `value`"}
        torch.cuda.empty_cache()

    # ---- how many distinct batches fit: raw signal per batch, plus ONE shared set of worst-case output slots, decoded
    # copy and library scratch (encode 9/8, decode 17/8 of the raw bytes, grown by 1/8 when (re)allocated)
    free = torch.cuda.mem_get_info(dev)[0]
    per_read = 200e3
    fixed = lambda k: k * per_read * (2.14 + 1.0 + 2.125 * 1.125) + 20e9  # noqa: E731
    if n == DEFAULT_READS:  # the default needs ~85 GB for one batch: step down on a GPU that does not have it free
        while n > 8192 and free < fixed(n) + n * per_read:
            n //= 2
    R = args.resident or max(1, min(args.steps, 16, int((free - fixed(n)) // (n * per_read * 1.02))))
    fixed_job = args.workload == "config5"

    # ---- the work queue: rank 0 owns the read table (lengths of world x R x n reads), every rank takes the contiguous,
    # sample-balanced range the partition gives it and cuts it into R batches (weak scaling: per-GPU work is fixed).
    # configs[4] (--workload config5) is a FIXED job instead: --total-reads reads for all ranks together (strong scaling);
    # a rank's share is cut into batches of at most n reads, all resident, each coded once in the timed region.
    total_reads = args.total_reads if fixed_job else world * R * n
    if rank == 0:
        lengths = torch.cat([codec.synth_lengths(5, f, min(1 << 20, total_reads - f)).to(torch.int64) for f in range(0, total_reads, 1 << 20)])
    else:
        lengths = torch.zeros(total_reads, dtype=torch.int64, device=dev)
    lengths = shard.share_read_table(lengths, coll_dev)
    my_first, my_last = shard.partition_reads(lengths, world)[rank]
    if fixed_job:
        R = max(1, -(-(my_last - my_first) // n))
        need = R * n * per_read * 1.02 + fixed(n)
        if need > free:
            raise SystemExit("bench.py --workload config5: this rank's share (%d reads, %.0f GB resident) does not fit %.0f GB of free HBM; "
                             "use more GPUs or fewer --total-reads" % (my_last - my_first, need / 1e9, free / 1e9))
        args.steps = R          # one pass over the rank's share
        args.warmup = min(args.warmup, 1)
    ranges = shard.cut_batches(my_first, my_last, lengths, R)

    max_total = max_ctotal = 0
    batches = []
    for (a, b) in ranges:
        lens = codec.synth_lengths(5, a, b - a)
        assert torch.equal(lens.cpu().to(torch.int64), lengths[a:b]), "read table mismatch between ranks"
        sizes = lens.to(torch.int64) * 2
        off, total = batch.layout(sizes.cpu(), 64)
        caps = max_compressed_sizes(sizes.cpu(), 1)
        coff, ctotal = batch.layout(caps, 64)
        raw = torch.zeros(total, dtype=torch.uint8, device=dev)  # zeros: the alignment gaps between reads compare equal below
        off = off.to(dev)
        codec.synth_signal(5, a, raw, off, lens)
        max_total, max_ctotal = max(max_total, total), max(max_ctotal, ctotal)
        batches.append(dict(first=a, n=b - a, raw=raw, off=off, size=sizes.to(torch.int32), coff=coff.to(dev), cap=caps.to(torch.int32).to(dev),
                            csize=torch.zeros(b - a, dtype=torch.int32, device=dev), raw_bytes=int(sizes.sum()), samples=int(lens.sum()),
                            total=total))
    for i in (0, len(batches[0]["cap"]) - 1):  # the vectorised bound is the library's
        assert int(batches[0]["cap"][i]) == L.vbz_max_compressed_size(int(batches[0]["size"][i]), ctypes.byref(opts))
    comp = torch.empty(max_ctotal, dtype=torch.uint8, device=dev)   # shared by all batches (outputs)
    back = torch.empty(max_total, dtype=torch.uint8, device=dev)
    res = torch.zeros(max(b["n"] for b in batches), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step(B, o=opts, mid=None):
        codec.compress(B["raw"], B["off"], B["size"], comp, B["coff"], B["cap"], B["csize"], o)
        if mid is not None:
            mid.record()
        codec.decompress(comp, B["coff"], B["csize"], back, B["off"], B["size"], res[: B["n"]], o)

    # ---- every resident batch round-trips, on the device (untimed; also the first warm-up)
    comp_bytes_all = 0
    for B in batches:
        back.zero_()  # the arena is shared by batches with different layouts: clear the previous batch's bytes in the gaps
        step(B)
        assert bool((res[: B["n"]] == B["size"]).all()), "decode failed for some read"
        assert torch.equal(B["raw"], back[: B["total"]]), "round trip mismatch"
        B["comp_bytes"] = int(B["csize"].to(torch.int64).sum())
        comp_bytes_all += B["comp_bytes"]
    for i in range(args.warmup):
        step(batches[i % R])
    torch.cuda.synchronize()

    codec.profile_reset()
    codec.profile(True)
    ev0 = torch.cuda.Event(enable_timing=True)
    evm = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    eve = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    raw_bytes = samples = comp_bytes = reads_done = 0
    ev0.record()
    for i in range(args.steps):
        B = batches[i % R]
        step(B, mid=evm[i])
        eve[i].record()
        raw_bytes += B["raw_bytes"]
        samples += B["samples"]
        comp_bytes += B["comp_bytes"]
        reads_done += B["n"]
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    codec.profile(False)
    # what every rank did, gathered: its distinct reads (the shares must add up to the job), the raw bytes it coded in the timed
    # region and the time that took it -- the line's `collective` object, so that a multi-GPU line certifies itself
    rows = shard.gather_row([sum(B["n"] for B in batches), raw_bytes, int(elapsed * 1e6)], coll_dev)
    assert int(rows[:, 0].sum()) == total_reads, "the ranks' shares do not add up to the job: %s of %d reads" % (rows[:, 0].tolist(), total_reads)
    coll = dict(shard.describe(backend, devices), reads_per_rank=rows[:, 0].tolist(),
                per_rank_MBps=[round(float(b_) / max(float(u_), 1.0), 1) for _, b_, u_ in rows.tolist()])
    elapsed = shard.max_over_ranks(elapsed, coll_dev)
    prof = codec.profile_read()
    ratio = raw_bytes / comp_bytes
    table, _ = shard.exchange_tallies(reads_done, raw_bytes, comp_bytes, coll_dev)
    total_raw = int(table[:, 1].sum())

    # encode / decode split from the events on torch's stream (the codec launches on it)
    enc_ms = dec_ms = 0.0
    prev = ev0
    for i in range(args.steps):
        enc_ms += prev.elapsed_time(evm[i])
        dec_ms += evm[i].elapsed_time(eve[i])
        prev = eve[i]

    if rank == 0:
        # ---- stage level: svb only (zstd level 0), same batch, a few steps
        stage = None
        if not args.no_stages:
            o0 = codec.options(True, 2, 0, 1)
            B = batches[0]
            back.zero_()
            step(B, o0)
            assert torch.equal(B["raw"], back[: B["total"]]), "svb-only round trip mismatch"
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            k = 5
            e[0].record()
            for _ in range(k):
                codec.compress(B["raw"], B["off"], B["size"], comp, B["coff"], B["cap"], B["csize"], o0)
            e[1].record()
            for _ in range(k):
                codec.decompress(comp, B["coff"], B["csize"], back, B["off"], B["size"], res[: B["n"]], o0)
            e[2].record()
            torch.cuda.synchronize()
            t_e, t_d = e[0].elapsed_time(e[1]) / k, e[1].elapsed_time(e[2]) / k
            s_bytes = int(B["csize"].to(torch.int64).sum())
            alg = B["raw_bytes"] + s_bytes  # per direction: 2 + s bytes per sample (SURVEY 8d "stage-level reporting")
            stage = {"svb_only": {
                "encode_MBps": round(B["raw_bytes"] / t_e / 1e3, 1), "decode_MBps": round(B["raw_bytes"] / t_d / 1e3, 1),
                "svb_bytes_per_sample": round(s_bytes / B["samples"], 4),
                "encode_hbm_frac": round(alg / (t_e * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "decode_hbm_frac": round(alg / (t_d * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "note": "zstd level 0 through the same batched entry points (byte-identical to the reference's level-0 output); "
                        "algorithmic bytes per direction = raw + svb stream"}}

        # ---- roofline: algorithmic bytes = (2 + c) per sample per direction, c = 2 / ratio.  `achieved` is quoted for the
        # direction that holds the dominant kernel (largest total time): that direction's algorithmic bytes over the WALL time of the
        # direction's launch sequence (HIP events on the codec's stream around the compress / the decompress call of every timed step).
        # Since round 6 a call runs its reads as two halves on two streams, so the library's per-label launch times overlap and their sum
        # is no longer a duration; the events around the call are.  Dividing by the dominant kernel alone would credit it with the other
        # kernels' work (reported as `dominant_kernel_alone`).
        per_launch = {k: v[1] / max(v[0], 1) * (v[0] / args.steps) for k, v in prof.items()}   # ms per step under each label (a label may be launched once per half)
        enc_k = [k for k in per_launch if k in ENCODE_LABELS]
        dec_k = [k for k in per_launch if k in DECODE_LABELS]
        spl = samples / args.steps  # samples per launch
        c = 2.0 / ratio
        alg_dir = (2.0 + c) * spl
        t_enc = enc_ms / args.steps * 1e-3
        t_dec = dec_ms / args.steps * 1e-3
        dom = max((k for k in per_launch if k in ENCODE_LABELS + DECODE_LABELS), key=lambda k: prof[k][1])
        dom_dir = "encode" if dom in enc_k else "decode"
        t_dom_dir = t_enc if dom_dir == "encode" else t_dec
        achieved = alg_dir / t_dom_dir / 1e9
        ktab, traffic_src, traffic_reads = profile_tables()
        scale = (n / traffic_reads) if traffic_reads else 1.0
        dir_labels = ENCODE_LABELS if dom_dir == "encode" else DECODE_LABELS

        def label_traffic(labels):
            rows = [v for v in ktab.values() if v["label"] in labels and v["hbm_bytes"] is not None]
            return int(sum(v["hbm_bytes"] for v in rows) * scale) if rows else None

        dir_traffic = label_traffic(dir_labels)
        if traffic_src and traffic_reads != n:
            traffic_src = "%s, scaled from %d to %d reads per launch" % (traffic_src, traffic_reads, n)
        # per kernel, from that committed profile: launch time, HBM bytes (PMC) and the HBM rate the kernel actually sustained
        per_kernel = {k: {"label": v["label"], "ms": round(v["ms"], 4), "hbm_GB": None if v["hbm_bytes"] is None else round(v["hbm_bytes"] / 1e9, 3),
                          "hbm_GBps": None if v["hbm_bytes"] is None or v["ms"] <= 0 else round(v["hbm_bytes"] / 1e9 / (v["ms"] * 1e-3), 1)}
                      for k, v in ktab.items() if v["label"] in ENCODE_LABELS + DECODE_LABELS and v["ms"] >= 0.02}
        roof = {
            "bound": "hbm",
            "kernel": dom,
            "direction": dom_dir,
            "achieved": round(achieved, 2),
            "peak": PEAK_HBM_GBS,
            "unit": "GB/s",
            "frac": round(achieved / PEAK_HBM_GBS, 5),
            "traffic": dir_traffic,
            "traffic_source": traffic_src and ("profiles/%s_hbm_traffic.csv: every kernel of the direction's labels (%s)" % (traffic_src, "+".join(dir_labels))),
            "traffic_over_algorithmic": None if not dir_traffic else round(dir_traffic / alg_dir, 3),
            "algorithmic_bytes_per_launch": int(alg_dir),
            "algorithmic_bytes_per_sample": round(2.0 + c, 4),
            "avg_launch_ms": round(t_dom_dir * 1e3, 4),
            "definition": "SURVEY 8d: (2 + c) bytes per int16 sample per direction; achieved = that x samples per call / wall time of the direction's "
                          "launch sequence (HIP events on the codec's stream around the %s call; its kernels: %s)"
                          % ("compress" if dom_dir == "encode" else "decompress", "+".join(dir_labels)),
            "per_direction": {
                "encode": {"achieved": round(alg_dir / t_enc / 1e9, 2), "frac": round(alg_dir / t_enc / 1e9 / PEAK_HBM_GBS, 5), "ms": round(t_enc * 1e3, 4),
                           "traffic": label_traffic(ENCODE_LABELS)},
                "decode": {"achieved": round(alg_dir / t_dec / 1e9, 2), "frac": round(alg_dir / t_dec / 1e9 / PEAK_HBM_GBS, 5), "ms": round(t_dec * 1e3, 4),
                           "traffic": label_traffic(DECODE_LABELS)},
            },
            "end_to_end": {"achieved": round(2 * alg_dir / (elapsed / args.steps) / 1e9, 2),
                           "frac": round(2 * alg_dir / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS, 5),
                           "note": "encode + decode algorithmic bytes over the wall time of a step"},
            "dominant_kernel_alone": {"achieved": round(alg_dir / (per_launch[dom] * 1e-3) / 1e9, 2),
                                      "note": "the literal per-label formula (the label's launches of a step, summed); flatters the kernel: its direction has other launches"},
            "per_kernel": per_kernel,
            "per_kernel_source": traffic_src and "profiles/%s_kernel_stats.csv + _hbm_traffic.csv (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE / WRITE_SIZE, gfx950 read correction)" % traffic_src.split(",")[0],
        }
        out = {
            "metric": METRIC,
            "value": round(total_raw / elapsed / 1e6, 1),
            "unit": "MB/s",
            "n_gpus": world,
            "collective": coll,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if fixed_job else "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {
                "workload": ("configs[4]: a FIXED job of %d reads (%.1f GB raw over all ranks) sharded over %d GPU(s) by cumulative samples, every rank's share "
                             "resident and coded once; per batch: " % (total_reads, total_raw / 1e9, world) if fixed_job else "") +
                            "configs[1]: synthetic int16 reads of ~100k samples (SURVEY 8d generator, seed 5), "
                            "~%d reads (%.2f GB raw) per step per GPU (batches are cut by cumulative samples), %d distinct batches resident per GPU (%d distinct reads over %d GPU(s), each "
                            "round-trip verified before the timed region), zig-zag + svb + zstd-format stage, encode then decode, inputs resident in HBM"
                            % (batches[0]["n"], batches[0]["raw_bytes"] / 1e9, R, total_reads, world),
                "reads_per_step": n,
                "reads_in_first_batch": batches[0]["n"],
                "distinct_reads": total_reads,
                "options": "zigzag=1,integer_size=2,zstd_level=1,vbz_version=1",
                "parallelism": "read table partitioned by cumulative samples across %d GPU(s) (rank 0 broadcasts the table, "
                               "all-gather of tallies); no data-path collective" % world,
            },
            "rank_imbalance": round(float(table[:, 1].max()) * world / max(float(table[:, 1].sum()), 1.0), 4),   # slowest rank's raw bytes over the mean
            "ratio": round(ratio, 4),
            "encode_MBps": round(raw_bytes / (enc_ms * 1e-3) / 1e6, 1),
            "decode_MBps": round(raw_bytes / (dec_ms * 1e-3) / 1e6, 1),
            "kernels_ms_per_launch": {k: round(v, 4) for k, v in per_launch.items()},   # per STEP under each label (summed over a call's halves: they overlap)
            "roofline": roof,
        }
        if stage:
            out["stages"] = stage
        if host_resident is not None:
            out["host_resident"] = host_resident
        if world == 1:   # (the legs below allocate: the resident batches have done their work)
            del comp, back
            for B in batches[1:]:
                B.clear()
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu and not fixed_job:
            out["decode_reference_frames"] = decode_reference_frames(codec)
        if world == 1 and not args.no_configs and not fixed_job:
            # BASELINE configs[0] and configs[3] in the same line: short runs of `--workload config1` / `config4`
            # (8 buffers per call, and one) on this process's codec; the full lines are what those workloads print on their own
            cfg = {}
            for key, wl, nb in (("config1", "config1", 1), ("config4", "config4", 8), ("config4_one_buffer", "config4", 1)):
                r = measure_large(wl, nb, 10, 2, codec, dev, rank, world, coll_dev, barrier, coll, cpu=(key == "config4" and not args.no_cpu), cpu_seconds=3.0)
                cfg[key] = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                            "encode_ms": r["encode_ms"], "decode_ms": r["decode_ms"], "ratio": r["ratio"],
                            "roofline_frac": r["roofline"]["frac"], "roofline_per_direction": r["roofline"]["per_direction"]}
                if "host_api" in r:
                    cfg[key]["host_api"] = {k: r["host_api"][k] for k in ("vbz_compress_ms", "vbz_decompress_ms")}
                if "cpu_baseline" in r:
                    cfg[key]["cpu_baseline"] = {k: r["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "one_thread", "ratio")}
                torch.cuda.empty_cache()
            out["configs"] = cfg
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
            if "single_socket" in out["cpu_baseline"]:
                out["vs_single_socket_cpu"] = round(out["value"] / out["cpu_baseline"]["single_socket"]["value"], 2)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()
    return 0


def run_large(args, codec, dev, rank, world, coll_dev, barrier, coll):
    out = measure_large(args.workload, args.buffers, args.steps, args.warmup, codec, dev, rank, world, coll_dev, barrier, coll, cpu=not args.no_cpu)
    if out is not None:
        print(json.dumps(out), flush=True)


def measure_large(workload, buffers, steps, warmup, codec, dev, rank, world, coll_dev, barrier, coll, cpu=True, cpu_seconds=6.0):
    """BASELINE.json configs[3] (`--workload config4`: uint32, no zig-zag, level 3, 10 M-element buffers) and configs[0]
    (`--workload config1`: one 400 k-sample int16 read): batches of few, large buffers, which the library spreads over many
    workgroups (segmented svb kernels, one wavefront per span of the entropy stage)."""
    import torch

    from vbz_compression_amd import _lib, batch, shard, vbz

    L = codec.L

    class A:   # (the body below was written against the argument namespace)
        pass

    args = A()
    args.workload, args.buffers, args.steps, args.warmup, args.no_cpu = workload, buffers, steps, warmup, not cpu
    if args.workload == "config4":
        elem, zz, level, ver, count, nbuf, kind = 4, False, 3, 0, 10_000_000, args.buffers, "u32"
        name = "configs[3]: uint32, no zig-zag, zstd level 3 (UD=32020,5,0,0,4,0,3), %d buffer(s) of 10M elements per step" % nbuf
    else:
        elem, zz, level, ver, count, nbuf, kind = 2, True, 1, 1, 400_000, 1, "i16"
        name = "configs[0]: ONE 400k-sample int16 read per step, zig-zag + svb + zstd-format stage"
    opts = codec.options(zz, elem, level, ver)
    nbytes = count * elem
    sizes = torch.full((nbuf,), nbytes, dtype=torch.int64)
    off, total = batch.layout(sizes, 64)
    cap = L.vbz_max_compressed_size(nbytes, ctypes.byref(opts))
    coff, ctotal = batch.layout(torch.full((nbuf,), cap, dtype=torch.int64), 64)
    raw = torch.zeros(total, dtype=torch.uint8, device=dev)
    offd, coffd = off.to(dev), coff.to(dev)
    lens = torch.full((nbuf,), count, dtype=torch.int32, device=dev)
    first = rank * nbuf
    (codec.synth_u32 if kind == "u32" else codec.synth_signal)(5, first, raw, offd, lens)
    comp = torch.zeros(ctotal, dtype=torch.uint8, device=dev)
    back = torch.zeros_like(raw)
    size32 = sizes.to(torch.int32).to(dev)
    cap32 = torch.full((nbuf,), cap, dtype=torch.int64).to(torch.int32).to(dev)
    csize = torch.zeros(nbuf, dtype=torch.int32, device=dev)
    res = torch.zeros(nbuf, dtype=torch.int32, device=dev)

    def step(mid=None):
        codec.compress(raw, offd, size32, comp, coffd, cap32, csize, opts)
        if mid is not None:
            mid.record()
        codec.decompress(comp, coffd, csize, back, offd, size32, res, opts)

    step()
    torch.cuda.synchronize()
    assert bool((res == size32).all()) and torch.equal(raw, back), "round trip mismatch"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    codec.profile_reset()
    codec.profile(True)
    ev0 = torch.cuda.Event(enable_timing=True)
    evm = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    eve = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(evm[i])
        eve[i].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, coll_dev)
    codec.profile(False)
    prof = codec.profile_read()
    raw_bytes = nbuf * nbytes * args.steps
    comp_bytes = int(csize.to(torch.int64).sum())
    ratio = nbuf * nbytes / comp_bytes
    table, _ = shard.exchange_tallies(nbuf * args.steps, raw_bytes, comp_bytes * args.steps, coll_dev)
    total_raw = int(table[:, 1].sum())
    enc_ms = dec_ms = 0.0
    prev = ev0
    for i in range(args.steps):
        enc_ms += prev.elapsed_time(evm[i])
        dec_ms += evm[i].elapsed_time(eve[i])
        prev = eve[i]
    if rank != 0:
        return None
    per_launch = {k: v[1] / max(v[0], 1) for k, v in prof.items()}
    # per direction: elem + c bytes per value ("for uint32 config: 4 + c4 per element per direction")
    c = elem / ratio
    alg_dir = (elem + c) * count * nbuf
    t_enc = enc_ms / args.steps * 1e-3
    t_dec = dec_ms / args.steps * 1e-3
    dom = max((k for k in per_launch if k not in ("plan_scratch", "seg_plan")), key=lambda k: prof[k][1])
    dom_dir = "encode" if "encode" in dom else "decode"
    t_dom = t_enc if dom_dir == "encode" else t_dec
    # HBM traffic of the dominant direction per step, from the newest committed PMC summary of this workload
    # (profiles/*_config4_hbm_traffic.csv: per-kernel bytes per launch x launches per step)
    traffic = traffic_src = None
    if kind == "u32":
        traffic, traffic_src = committed_traffic_large(dom_dir, nbuf)
    out = {
        "metric": METRIC if kind != "u32" else "MB/s encode+decode, uint32 buffers (BASELINE configs[3]: UD=32020,5,0,0,4,0,3), 1 MI355X vs CPU; ratio preserved",
        "value": round(total_raw / elapsed / 1e6, 1), "unit": "MB/s", "n_gpus": world, "collective": coll, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "uint32" if kind == "u32" else "int16", "data": "synthetic",
        "config": {"workload": name + ", encode then decode through the batched entry points, inputs resident in HBM",
                   "buffers_per_step": nbuf, "elements_per_buffer": count,
                   "options": "zigzag=%d,integer_size=%d,zstd_level=%d,vbz_version=%d" % (zz, elem, level, ver),
                   "parallelism": "%d GPU(s), the same workload on each (buffers are independent; no data-path collective)" % world},
        "ratio": round(ratio, 4),
        "encode_MBps": round(raw_bytes / (enc_ms * 1e-3) / 1e6, 1), "decode_MBps": round(raw_bytes / (dec_ms * 1e-3) / 1e6, 1),
        "encode_ms": round(enc_ms / args.steps, 4), "decode_ms": round(dec_ms / args.steps, 4),
        "kernels_ms_per_launch": {k: round(v, 4) for k, v in per_launch.items()},
        "roofline": {"bound": "hbm", "kernel": dom, "direction": dom_dir, "achieved": round(alg_dir / t_dom / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(alg_dir / t_dom / 1e9 / PEAK_HBM_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": int(alg_dir), "algorithmic_bytes_per_value": round(elem + c, 4), "avg_launch_ms": round(t_dom * 1e3, 4),
                     "definition": "SURVEY 8d: (integer size + c) bytes per value per direction over the duration of the direction's launch "
                                   "sequence (HIP events on the codec's stream around every call); a batch of few buffers is bound by the latency "
                                   "of one wavefront per span, not by HBM",
                     "per_direction": {"encode": {"achieved": round(alg_dir / t_enc / 1e9, 2), "frac": round(alg_dir / t_enc / 1e9 / PEAK_HBM_GBS, 5)},
                                       "decode": {"achieved": round(alg_dir / t_dec / 1e9, 2), "frac": round(alg_dir / t_dec / 1e9 / PEAK_HBM_GBS, 5)}}},
    }
    if world == 1:
        # the single-buffer host API (vbz_compress / vbz_decompress: PCIe copies and synchronisation included; never `value`)
        h = raw[:nbytes].cpu().numpy()
        o2 = _lib.CompressionOptions(zz, elem, level, ver)
        f = vbz.compress_raw(h, o2)
        b2 = vbz.decompress_raw(f, nbytes, o2)
        assert b2.tobytes() == h.tobytes()
        # the C entry points themselves, on buffers the caller owns and reuses (what a C caller does; the numpy wrappers above
        # allocate a fresh 40 MB array per call, whose page faults would be most of the time measured)
        import numpy as np
        bound = L.vbz_max_compressed_size(nbytes, ctypes.byref(o2))
        cbuf = np.zeros(bound + 16, np.uint8)
        dbuf = np.zeros(nbytes, np.uint8)
        k = 10
        n = 0
        for it in range(k + 2):
            if it == 2:
                tc0 = time.perf_counter()
            n = L.vbz_compress(h.ctypes.data, nbytes, cbuf.ctypes.data, bound, ctypes.byref(o2))
        tc1 = time.perf_counter()
        assert not _lib.is_error(n) and n == len(f)
        for it in range(k + 2):
            if it == 2:
                td0 = time.perf_counter()
            m = L.vbz_decompress(cbuf.ctypes.data, n, dbuf.ctypes.data, nbytes, ctypes.byref(o2))
        td1 = time.perf_counter()
        assert m == nbytes and dbuf.tobytes() == h.tobytes()
        out["host_api"] = {"vbz_compress_ms": round((tc1 - tc0) / k * 1e3, 3), "vbz_decompress_ms": round((td1 - td0) / k * 1e3, 3),
                           "encode_decode_MBps": round(nbytes / (((tc1 - tc0) + (td1 - td0)) / k) / 1e6, 1),
                           "note": "one buffer per call through include/vbz.h, pageable host memory in and out, buffers reused by the caller"}
    if world == 1 and not args.no_cpu:
        if kind == "u32":
            out["cpu_baseline"] = cpu_baseline_u32(max(nbuf, usable_cpus()[0]), count, min_seconds=cpu_seconds)
        else:
            out["cpu_baseline"] = cpu_baseline(min_seconds=4.0, n_reads=16384)
            out["cpu_baseline"]["note"] = "reads of ~100k samples of the same generator (a single 400k read is one core's work: see one_thread)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=DEFAULT_READS, help="reads per batch (one batch per step)")
    ap.add_argument("--resident", type=int, default=0, help="distinct batches kept in HBM and cycled (0: as many as fit, at most 16)")
    ap.add_argument("--workload", default="reads", choices=["reads", "config4", "config1", "config5"],
                    help="reads: BASELINE configs[1] (the headline); config4: uint32 10M-element buffers; config1: one 400k-sample read; "
                         "config5: BASELINE configs[4], a FIXED job of --total-reads reads (~100 GB) sharded over the ranks (strong scaling)")
    ap.add_argument("--total-reads", type=int, default=500000, help="config5: reads of the whole job (500 000 x ~100 k samples = 100 GB raw)")
    ap.add_argument("--buffers", type=int, default=8, help="config4: buffers per step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the work queue (nccl = RCCL)")
    ap.add_argument("--dry-run", action="store_true", help="launcher and work-queue plumbing only: no codec, no GPU (CPU test of the N > 1 path)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-resident (PCIe-inclusive) leg")
    ap.add_argument("--no-stages", action="store_true", help="skip the svb-only stage line")
    ap.add_argument("--no-configs", action="store_true", help="skip the short configs[0] / configs[3] runs of the default line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.dry_run:
            # the parent must not touch the HIP runtime (its children would be forks of an initialised process): the GPUs are
            # counted from the kernel driver's topology; when that cannot be read the ranks find out for themselves
            have = kfd_gpu_count()
            if have is not None and have < args.gpus:
                raise SystemExit("bench.py: --gpus %d but %d GPU(s) in /sys/class/kfd" % (args.gpus, have))
        return launch_ranks(args, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
