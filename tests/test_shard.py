"""CPU tests of the multi-GPU plumbing (vbz_compression_amd/shard.py) with the gloo backend, world_size 2."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vbz_compression_amd import shard


def test_partition_reads_is_contiguous_and_balanced():
    g = torch.Generator().manual_seed(1)
    lengths = torch.randint(90000, 110001, (1000,), generator=g)
    for world in (1, 2, 3, 8):
        parts = shard.partition_reads(lengths, world)
        assert parts[0][0] == 0 and parts[-1][1] == 1000
        for (a, b), (c, d) in zip(parts, parts[1:]):
            assert b == c and a <= b
        loads = [int(lengths[a:b].sum()) for a, b in parts]
        assert max(loads) - min(loads) <= 2 * 110000
    assert shard.partition_reads([], 4) == [(0, 0)] * 4


def test_skewed_read_table_balances_within_two_percent():
    """Real files hold reads of 9 885 ... 505 057 samples: a table with that spread, sorted the
    worst way (long reads first), still ends within 2 % load balance, and a rank's batches are balanced the same way."""
    g = torch.Generator().manual_seed(7)
    lengths = torch.cat([torch.randint(300000, 505058, (3000,), generator=g), torch.randint(9885, 40000, (29000,), generator=g)])
    for world in (2, 4, 8):
        parts = shard.partition_reads(lengths, world)
        loads = [int(lengths[a:b].sum()) for a, b in parts]
        assert (max(loads) - min(loads)) / (sum(loads) / world) < 0.02, loads
        a, b = parts[-1]
        cuts = shard.cut_batches(a, b, lengths, 4)
        assert cuts[0][0] == a and cuts[-1][1] == b and all(x[1] == y[0] for x, y in zip(cuts, cuts[1:]))
        bl = [int(lengths[x:y].sum()) for x, y in cuts]
        assert (max(bl) - min(bl)) / (sum(bl) / 4) < 0.02


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table, offset = shard.exchange_tallies(reads=10 + rank, raw_bytes=1000 * (rank + 1), compressed_bytes=400 + 50 * rank)
        mx = shard.max_over_ranks(1.5 + rank)
        # rank 0 owns the read table; everybody derives the same partition from the broadcast copy
        lengths = torch.arange(1000, 1100, dtype=torch.int64) if rank == 0 else torch.zeros(100, dtype=torch.int64)
        lengths = shard.share_read_table(lengths)
        plan = shard.partition_reads(lengths, world)
        out.put((rank, table.tolist(), offset, mx, plan, int(lengths.sum())))
    finally:
        dist.destroy_process_group()


def test_work_queue_metadata_exchange_world_size_2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, t0, off0, mx0, plan0, sum0), (r1, t1, off1, mx1, plan1, sum1) = res
    assert t0 == t1 == [[10, 1000, 400], [11, 2000, 450]]
    assert (off0, off1) == (0, 400)          # exclusive scan of compressed bytes = global output offsets
    assert mx0 == mx1 == 2.5                  # MAX over ranks (the timed region of bench.py)
    assert sum0 == sum1 == sum(range(1000, 1100))
    assert plan0 == plan1 and plan0[0][0] == 0 and plan0[0][1] == plan0[1][0] and plan0[1][1] == 100


def test_bench_launcher_starts_one_process_per_rank():
    """`python bench.py --gpus 2` with no WORLD_SIZE is a launcher: it starts two rank processes which form a process
    group (gloo here, RCCL on GPUs) and run the same work-queue code as the real benchmark; --dry-run leaves the codec
    out (there is no GPU in this test).  n_gpus, the sample-balanced ranges and the all-gathered tallies are checked."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--reads", "300", "--steps", "4"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    (a0, b0), (a1, b1) = out["ranges"]
    assert a0 == 0 and b0 == a1 and b1 == 2 * 2 * 300
    t = out["tallies"]
    assert t[0][0] == b0 - a0 and t[1][0] == b1 - a1
    assert abs(t[0][1] - t[1][1]) / t[0][1] < 0.01      # balanced by samples
    # the line certifies its own collective: the backend, the world size the process group reports, every rank's device ordinal
    # as gathered over it, what every rank did (the shares add up to the job: bench.py asserts it in every rank) and its rate
    c = out["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and sorted(c["devices"]) == [0, 1]
    assert c["reads_per_rank"] == [b0 - a0, b1 - a1] and sum(c["reads_per_rank"]) == 2 * 2 * 300
    assert len(c["per_rank_MBps"]) == 2 and all(v > 0 for v in c["per_rank_MBps"])
    # a world size that does not match --gpus is refused, and a failing rank makes the launcher fail
    env2 = dict(env, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=120, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_fixed_job_is_sharded_over_the_ranks():
    """BASELINE configs[4] (`--workload config5`): a FIXED number of reads for the whole job, cut over the ranks by cumulative
    samples (strong scaling), every rank coding its share in batches of at most --reads reads.  Dry run, two ranks, gloo."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    outs = {}
    for gpus in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--workload", "config5", "--dry-run", "--total-reads",
                            "3001", "--reads", "500"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        outs[gpus] = json.loads(lines[0])
    one, two = outs[1], outs[2]
    assert one["scaling"] == two["scaling"] == "strong" and two["n_gpus"] == 2
    assert one["ranges"] == [[0, 3001]] and one["steps"] == 7                 # 3001 reads in batches of at most 500
    (a0, b0), (a1, b1) = two["ranges"]
    assert a0 == 0 and b0 == a1 and b1 == 3001                                # the same job, not a bigger one
    assert two["steps"] in (3, 4)                                             # rank 0: ~1500 reads in batches of at most 500
    t = two["tallies"]
    assert t[0][0] + t[1][0] == 3001 and t[0][1] + t[1][1] == one["tallies"][0][1]
    assert 1.0 <= two["rank_imbalance"] < 1.01
    assert one["collective"] == dict(one["collective"], backend=None, world_size=1, devices=[0], reads_per_rank=[3001])
    assert two["collective"]["world_size"] == 2 and sum(two["collective"]["reads_per_rank"]) == 3001


def test_fixed_job_of_half_a_million_reads_over_eight_ranks():
    """BASELINE configs[4] at its stated size -- 500 000 reads of ~100 k samples, 100 GB -- over EIGHT ranks: launcher, process group
    (gloo here, RCCL on the node), read-table broadcast, partition by cumulative samples, tallies.  No codec (dry run): what the day
    an 8-GPU node is there must already be right is the plumbing -- shares that add up to the job, ranks within 1 % of each other,
    contiguous ranges, one device per rank.  No 1 -> 8 curve has been measured on hardware (README)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "config5", "--dry-run", "--total-reads", "500000"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong"
    ranges = out["ranges"]
    assert len(ranges) == 8 and ranges[0][0] == 0 and ranges[-1][1] == 500000
    assert all(ranges[k][1] == ranges[k + 1][0] for k in range(7))
    c = out["collective"]
    assert c["world_size"] == 8 and sorted(c["devices"]) == list(range(8))
    assert sum(c["reads_per_rank"]) == 500000 and c["reads_per_rank"] == [b - a for a, b in ranges]
    assert 1.0 <= out["rank_imbalance"] <= 1.01
    t = out["tallies"]
    assert sum(row[0] for row in t) == 500000
    assert 0.95e11 < sum(row[1] for row in t) < 1.05e11          # ~100 GB of raw signal


def test_launcher_relays_the_first_failing_rank():
    """A rank that fails takes the others down with it: the launcher must fail too and show THAT rank's last words (every rank's stderr
    is kept in a file of its own)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # without --dry-run every rank needs a GPU: there is none here, so each says so and exits
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--reads", "64", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=dict(env, HIP_VISIBLE_DEVICES="-1"))
    if r.returncode == 0:
        pytest.skip("this box has GPUs: the ranks ran")
    if "GPU(s) in /sys/class/kfd" in r.stderr:
        pytest.skip("fewer GPUs than ranks: the launcher refused before it started any")
    assert "first to fail" in r.stderr and "rank exit codes" in r.stderr


def test_launcher_counts_gpus_without_the_hip_runtime():
    """The launcher parent must not initialise HIP (its children would be forks of an initialised process): it counts the
    GPUs from /sys/class/kfd, or not at all.  Here (no GPU, usually no kfd) the function answers None or a number and
    bench.py's parent path imports no torch.cuda call."""
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.kfd_gpu_count()
    assert n is None or (isinstance(n, int) and n >= 0)
    src = open(os.path.join(root, "bench.py")).read()
    launcher = src[src.index("def main():"):]
    assert "device_count" not in launcher and "is_available" not in launcher


def test_gather_row_and_describe_without_a_group():
    assert shard.gather_row([3, 4, 5]).tolist() == [[3, 4, 5]]
    assert shard.describe("nccl", [0]) == {"backend": None, "world_size": 1, "devices": [0]}


def test_single_rank_identity():
    table, off = shard.exchange_tallies(5, 100, 40)
    assert table.tolist() == [[5, 100, 40]] and off == 0
    assert shard.max_over_ranks(3.0) == 3.0
