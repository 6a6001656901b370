"""Read sharding across the GPUs of one node.

Reads are independent (one frame per read, no cross-read state: SURVEY.md section 8e), so the data
path has NO collective: every rank encodes/decodes its own contiguous range of the read table.
torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU) carries only work-queue
metadata: the per-rank tallies that give every rank its global output offset.
"""
import torch
import torch.distributed as dist


def partition_reads(lengths, world_size):
    """Split the read table into `world_size` contiguous ranges balanced by cumulative sample count.
    Returns a list of (first, last_exclusive) per rank.  Deterministic, identical on every rank."""
    lengths = torch.as_tensor(lengths, dtype=torch.int64)
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


def batch_plan(n_batches, rank, world_size):
    """Static round-robin work queue over batches of reads: rank r owns batches r, r+W, r+2W, ..."""
    return list(range(rank, n_batches, world_size))


def exchange_tallies(reads, raw_bytes, compressed_bytes, device=None):
    """All-gather {reads, raw_bytes, compressed_bytes} (3 x int64 per rank) and return
    (table[world, 3], my exclusive output offset in compressed bytes).  Without an initialised
    process group this is the single-rank identity."""
    mine = torch.tensor([int(reads), int(raw_bytes), int(compressed_bytes)], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return mine.unsqueeze(0).cpu(), 0
    world = dist.get_world_size()
    table = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(table, mine)
    table = torch.stack(table).cpu()
    offset = int(table[: dist.get_rank(), 2].sum())
    return table, offset


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (used for the timed region of the benchmark)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
