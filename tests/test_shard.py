"""CPU tests of the multi-GPU plumbing (vbz_compression_amd/shard.py) with the gloo backend, world_size 2."""
import os
import socket

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
    assert shard.batch_plan(10, 1, 4) == [1, 5, 9]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table, offset = shard.exchange_tallies(reads=10 + rank, raw_bytes=1000 * (rank + 1), compressed_bytes=400 + 50 * rank)
        mx = shard.max_over_ranks(1.5 + rank)
        plan = shard.batch_plan(7, rank, world)
        out.put((rank, table.tolist(), offset, mx, plan))
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
    (r0, t0, off0, mx0, plan0), (r1, t1, off1, mx1, plan1) = res
    assert t0 == t1 == [[10, 1000, 400], [11, 2000, 450]]
    assert (off0, off1) == (0, 400)          # exclusive scan of compressed bytes = global output offsets
    assert mx0 == mx1 == 2.5                  # MAX over ranks (the timed region of bench.py)
    assert plan0 == [0, 2, 4, 6] and plan1 == [1, 3, 5]


def test_single_rank_identity():
    table, off = shard.exchange_tallies(5, 100, 40)
    assert table.tolist() == [[5, 100, 40]] and off == 0
    assert shard.max_over_ranks(3.0) == 3.0
