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


def row_range(num_cells, rank, world, align=1):
    """Rows [lo, hi) of the matrix that `rank` keeps when the matrix stays sharded (contiguous, as even
    as possible). With align = the tile edge the cuts fall on cell-block boundaries, so that no cell
    block (and none of its tiles) is shared by two ranks."""
    units = (num_cells + align - 1) // align
    per, extra = divmod(units, world)
    lo = rank * per + min(rank, extra)
    hi = lo + per + (1 if rank < extra else 0)
    return min(lo * align, num_cells), min(hi * align, num_cells)


def sharded_rows(plan, acc, mutation_rate, homozygous_rate, seq_error_rate, rank, world, normalization="ADD_MIN",
                 group=None):
    """This rank's row block of the normalised matrix with nothing gathered (BASELINE config 5): the rank
    accumulates every tile that touches its rows (an off-diagonal tile is computed by the two ranks that
    own its row block and its column block: twice the pair work in total, no tile ever travels), the
    ranks agree on the maximum that ADD_MIN / SCALE_MAX_1 need with one scalar all-reduce, and each
    normalises its rows (cut on cell-block boundaries). `acc` from plan.new_acc(), zeroed here.
    Returns (rows tensor, row_begin)."""
    import torch
    import torch.distributed as dist
    lo, hi = row_range(plan.num_cells, rank, world, plan.block_cells)
    ids = plan.tiles_of_rows(lo, hi)
    acc.zero_()
    plan.accumulate_list(acc, mutation_rate, homozygous_rate, seq_error_rate, ids)
    local_max = plan.max_of_tiles(acc, ids)
    if world > 1:
        t = torch.tensor([local_max], dtype=torch.float64,
                         device=acc.device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        local_max = float(t.item())
    return plan.finalize_rows_max(acc, lo, hi, local_max, normalization), lo


class _DevicePointer:
    """A device buffer handed to the all-reduce callback, seen through the CUDA array interface."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2}


def sharded_eigenpairs(rows, row_begin, num_cells, n_values=20, n_vectors=7, tol=0.0, max_cycles=0, group=None):
    """Spectral step on a matrix kept sharded by rows (BASELINE config 5; reference
    spectral_clustering.cpp:127-138 on the whole matrix): every rank passes the (rows x num_cells)
    float64 CUDA tensor of its row block (`SimilarityMatrixPlan.finalize_rows`), the n x 32 partial
    products are summed with `torch.distributed.all_reduce` (RCCL on "nccl"; rehearsal backends such as
    gloo are staged through the host). Returns (eigenvalues, eigenvectors (num_cells x n_vectors CUDA
    tensor, the same on every rank), info) like `secedo_amd.smallest_eigenpairs`."""
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    from . import _lib
    assert rows.is_cuda and rows.dtype == torch.float64 and rows.is_contiguous()
    n_rows = rows.shape[0]
    assert n_rows == 0 or rows.shape[1] == num_cells
    direct = dist.get_backend(group) == "nccl"
    failure = []

    def allreduce(_ctx, ptr, count, _stream):
        try:
            buf = torch.as_tensor(_DevicePointer(ptr, count), device=rows.device)
            if direct:
                dist.all_reduce(buf, group=group)
            else:
                host = buf.cpu()
                dist.all_reduce(host, group=group)
                buf.copy_(host)
            return 0
        except Exception as e:  # an exception must not cross the C frames above
            failure.append(e)
            return 1

    hook = _lib.ALLREDUCE_SUM_FN(allreduce)
    n_values = min(n_values, num_cells, 32)
    n_vectors = min(n_vectors, n_values)
    vals = np.empty(n_values, dtype=np.float64)
    vecs = torch.empty((max(n_vectors, 1), num_cells), dtype=torch.float64, device=rows.device)
    info = _lib.SpectralInfo()
    stream = torch.cuda.current_stream(rows.device).cuda_stream
    rc = _lib.lib().secedo_spectral_eigs_rows_device(
        rows.device.index or 0, rows.data_ptr() if n_rows else None, row_begin, n_rows, num_cells, n_values,
        n_vectors, tol, max_cycles, _lib.ptr(vals), vecs.data_ptr(), C.byref(info), hook, None, stream)
    if failure:
        raise failure[0]
    _lib.check(rc)
    return vals, vecs[:n_vectors].T, {"cycles": info.cycles, "block_products": info.block_products,
                                      "converged": bool(info.converged),
                                      "max_residual_vectors": info.max_residual_vectors,
                                      "max_residual_values": info.max_residual_values}
