"""Read sharding across the GPUs of one node: the work queue of the VBZ path.

Reads are independent (one frame per read, no cross-read state: SURVEY.md section 8e), so the data
path has NO collective: every rank encodes/decodes its own contiguous range of the read table.
torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU) carries only work-queue
metadata: the read-length table (broadcast from rank 0, which owns it), from which every rank derives the
same sample-balanced partition, and the per-rank tallies that give every rank its global output offset.
Real files hold reads of 9 885 ... 505 057 samples, so the partition balances
cumulative samples, not read counts.
"""
import torch
import torch.distributed as dist


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def partition_reads(lengths, world_size):
    """Split the read table into `world_size` contiguous ranges balanced by cumulative sample count.
    Returns a list of (first, last_exclusive) per rank.  Deterministic, identical on every rank."""
    lengths = torch.as_tensor(lengths, dtype=torch.int64).cpu()
    n = int(lengths.numel())
    if n == 0:
        return [(0, 0)] * world_size
    csum = torch.cumsum(lengths, 0)
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        idx = int(torch.searchsorted(csum, torch.tensor(target, dtype=torch.int64), right=False))
        # the cut goes after the read that crosses the target if that is closer
        bounds.append(min(max(idx + (1 if idx < n and (int(csum[idx]) - target) <= int(lengths[idx]) // 2 else 0), bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def cut_batches(first, last, lengths, n_batches):
    """Cut one rank's range [first, last) of the read table into `n_batches` contiguous batches of near-equal
    sample count (the unit one vbz_gpu_*_batch call handles).  Returns [(first, last_exclusive), ...]."""
    lengths = torch.as_tensor(lengths, dtype=torch.int64).cpu()
    parts = partition_reads(lengths[first:last], n_batches)
    return [(first + a, first + b) for a, b in parts]


def share_read_table(lengths, device=None):
    """Rank 0 owns the read-length table; every other rank receives it (one broadcast, <= 4 MB for 1 M reads).
    `lengths` must have the same shape on every rank (contents matter on rank 0 only).  Returns an int64 CPU tensor."""
    t = torch.as_tensor(lengths, dtype=torch.int64)
    if not _active():
        return t.cpu()
    t = t.to(device) if device is not None else t.clone()
    dist.broadcast(t, src=0)
    return t.cpu()


def exchange_tallies(reads, raw_bytes, compressed_bytes, device=None):
    """All-gather {reads, raw_bytes, compressed_bytes} (3 x int64 per rank) and return
    (table[world, 3], my exclusive output offset in compressed bytes).  Without an initialised
    process group this is the single-rank identity."""
    mine = torch.tensor([int(reads), int(raw_bytes), int(compressed_bytes)], dtype=torch.int64, device=device)
    if not _active():
        return mine.unsqueeze(0).cpu(), 0
    world = dist.get_world_size()
    table = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(table, mine)
    table = torch.stack(table).cpu()
    offset = int(table[: dist.get_rank(), 2].sum())
    return table, offset


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (used for the timed region of the benchmark)."""
    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_row(values, device=None):
    """All-gather one row of int64 values per rank; returns table[world, len(values)] on the CPU (the single-rank identity
    without an initialised process group).  Work-queue metadata only: a few words per rank."""
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if not _active():
        return mine.unsqueeze(0).cpu()
    table = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(table, mine)
    return torch.stack(table).cpu()


def describe(backend, device_ordinals):
    """What carried the work queue, as the ranks themselves see it (bench.py's `collective` object): the backend, the world size the
    process group reports (1 without a group), and every rank's device ordinal as gathered over that group."""
    ws = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return {"backend": backend if ws > 1 else None, "world_size": ws, "devices": [int(d) for d in device_ordinals]}
