"""Row-range sharding of a table scan across the GPUs of one node.

Every output row depends only on its own input row and on read-only weights, so the scan
partitions by row range with NO data-path collective: rank r of W owns rows
[r*rows_per_rank, (r+1)*rows_per_rank) of the (W*rows_per_rank)-row table (weak scaling), weights
are replicated at infera_load_model time, and results are placed by row offset.  The only
cross-rank traffic is the control plane of the benchmark (a barrier and a max over ranks of one
scalar), carried by torch.distributed/gloo.
"""
from __future__ import annotations

from typing import Tuple


def row_range(rank: int, world: int, rows_per_rank: int) -> Tuple[int, int]:
    """[row0, row1) of the global table owned by `rank` (weak scaling: fixed rows per rank)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return rank * rows_per_rank, (rank + 1) * rows_per_rank


def split_rows(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Strong-scaling split of `total_rows` into `world` contiguous, near-equal ranges (the first
    total_rows % world ranks get one extra row)."""
    base, extra = divmod(total_rows, world)
    row0 = rank * base + min(rank, extra)
    return row0, row0 + base + (1 if rank < extra else 0)


def chunk_owner(chunk_index: int, n_devices: int) -> int:
    """Round-robin placement of DuckDB DataChunks onto devices inside one process
    (north_star: "partition DataChunks round-robin across the 8 GPUs")."""
    return chunk_index % n_devices


def max_over_ranks(value: float) -> float:
    """Max of one scalar over all ranks (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj):
    """[rank 0's obj, rank 1's obj, ...] on every rank (control plane: a few hundred bytes per rank); [obj] without a process group."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def barrier() -> None:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()
