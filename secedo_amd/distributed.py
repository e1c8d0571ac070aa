"""Multi-GPU driver of the similarity-matrix path: one process per GPU, RCCL over xGMI.

The output tiles (upper-triangular cell-block pairs) are independent, so they are dealt to the
ranks in contiguous ranges; the pileup is small (<= 0.4 GB packed) and is replicated. The only
exchange is one all-gather of the tile-major int64 accumulator, after which
every rank normalises and mirrors the full matrix locally -- integer accumulators make the result
bit-identical for any number of ranks.
"""
from __future__ import annotations

from typing import Tuple


def tiles_per_rank(num_tiles: int, world: int) -> int:
    return -(-num_tiles // world)


def tile_range(num_tiles: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of upper-triangular tiles owned by `rank` (may be empty)."""
    per = tiles_per_rank(num_tiles, world)
    return min(rank * per, num_tiles), min((rank + 1) * per, num_tiles)


def sharded_accumulate(plan, acc, mutation_rate, homozygous_rate, seq_error_rate, rank, world,
                       group=None):
    """Accumulate this rank's tiles into `acc` and all-gather the other ranks' tiles.

    `acc` must come from plan.new_acc(pad_tiles_to=world): every rank then owns an equally sized
    slice. The rank's slice is copied to a send buffer first: the collective's input and output do not
    alias (1/world of the accumulator, negligible next to the gather itself).
    """
    import torch.distributed as dist

    b2 = plan.block_cells ** 2
    per = tiles_per_rank(plan.num_tiles, world)
    assert acc.numel() == per * world * b2, "acc must be padded with new_acc(pad_tiles_to=world)"
    lo, hi = tile_range(plan.num_tiles, rank, world)
    mine = acc[rank * per * b2:(rank + 1) * per * b2]
    mine.zero_()
    plan.accumulate(acc, mutation_rate, homozygous_rate, seq_error_rate, lo, hi)
    if world > 1:
        if acc.is_cuda and dist.get_backend(group) != "nccl":
            # rehearsal backends (gloo) move host memory: stage the slices through the CPU
            gathered = acc.new_empty(acc.shape, device="cpu")
            dist.all_gather_into_tensor(gathered, mine.cpu(), group=group)
            acc.copy_(gathered)
        else:
            dist.all_gather_into_tensor(acc, mine.clone(), group=group)
    return acc
