"""Multi-GPU driver of the similarity-matrix path: one process per GPU, RCCL over xGMI.

Two ways to share one matrix among the ranks, both exact (integer accumulators make the result
bit-identical for any number of ranks):

* by output tiles (sharded_accumulate): the upper-triangular cell-block pairs are independent, so they
  are dealt to the ranks in contiguous ranges; the pileup is replicated and packed by every rank; one
  all-gather of the tile-major int64 accumulator.
* by chromosomes (chromosome_shard + chromosome_sharded_accumulate): reads, flushes and the tail rule
  never cross a chromosome (reference similarity_matrix.cpp:345-408 walks the chromosomes one after the
  other and empties its read table after each, :407-408; the `completed` counter it carries over, :344,
  is below the flush threshold there and is overtaken by the real count before it can trigger a flush:
  completed = max(carried, complete reads); tests/test_distributed_cpu.py checks the sum against a CPU
  restatement of the reference that carries it), so the sum over loci splits at chromosome boundaries: every rank packs and accumulates ONLY its chromosomes -- the packing, which is two
  thirds of a 1000-cell step, is divided too -- for all tiles, and one all-reduce (sum) of the
  accumulator replaces the all-gather. bench.py uses this one when there are chromosomes enough.

Either way every rank then normalises and mirrors the full matrix locally.
"""
from __future__ import annotations

from typing import Tuple


def _exchanges(world: int) -> bool:
    """Whether the collectives are issued: always with more than one rank; with ONE rank only when a process
    group exists and SECEDO_DIST_EXCHANGE_ALWAYS=1 -- a world-size-1 RCCL group on one GPU then runs every
    collective branch of this module (in-place all-gather, side-stream chunks, all-reduces) as the N-rank job
    does, so the one-GPU test box executes the RCCL code paths before an 8-GPU run does
    (tests/test_gpu_distributed.py::test_rccl_world_size_1_runs_every_collective_branch)."""
    if world > 1:
        return True
    import os
    if os.environ.get("SECEDO_DIST_EXCHANGE_ALWAYS", "") not in ("", "0"):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    return False


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
    plan.accumulate(acc, mutation_rate, homozygous_rate, seq_error_rate, lo, hi, overwrite=True)
    if hi - lo < per:
        acc[hi * b2:(rank + 1) * per * b2].zero_()  # the padding tiles of the last ranks' slices
    if _exchanges(world):
        if acc.is_cuda and dist.get_backend(group) != "nccl":
            # rehearsal backends (gloo) move host memory: stage the slices through the CPU
            gathered = acc.new_empty(acc.shape, device="cpu")
            dist.all_gather_into_tensor(gathered, mine.cpu(), group=group)
            acc.copy_(gathered)
        else:
            dist.all_gather_into_tensor(acc, mine.clone(), group=group)
    return acc


def sharded_accumulate_overlapped(plan, acc, mutation_rate, homozygous_rate, seq_error_rate, rank, world,
                                  chunks=4, group=None, comm_stream=None):
    """sharded_accumulate with the exchange hidden behind the accumulation: the rank's tile range is worked off
    in `chunks` launches, and as soon as a chunk's tiles are final (correct_tiles has stored them) its
    all-gather is issued on a second stream while the next chunk accumulates; the calling stream waits for the
    last exchange at the end. Tiles keep their place in `acc` (position = tile index), so chunk k of rank r is
    the k-th part of r's slice and the gather of chunk k writes `world` separate slices of `acc` (all_gather
    with a list of views). Bitwise the same accumulator as sharded_accumulate. `acc` from
    plan.new_acc(pad_tiles_to=world). On rehearsal backends (gloo) the chunks are exchanged one after the other
    through the host."""
    import torch
    import torch.distributed as dist

    b2 = plan.block_cells ** 2
    per = tiles_per_rank(plan.num_tiles, world)
    assert acc.numel() == per * world * b2, "acc must be padded with new_acc(pad_tiles_to=world)"
    lo, hi = tile_range(plan.num_tiles, rank, world)
    n_mine = hi - lo
    chunks = max(1, min(chunks, per))
    step = -(-per // chunks)
    exchange = _exchanges(world)
    direct = exchange and acc.is_cuda and dist.get_backend(group) == "nccl"
    main = torch.cuda.current_stream(acc.device) if acc.is_cuda else None
    if direct and comm_stream is None:
        comm_stream = torch.cuda.Stream(acc.device)
    for c_lo in range(0, per, step):
        c_hi = min(c_lo + step, per)  # the chunk, in tiles of a rank's slice
        t_lo, t_hi = lo + min(c_lo, n_mine), lo + min(c_hi, n_mine)
        if t_hi > t_lo:
            plan.accumulate(acc, mutation_rate, homozygous_rate, seq_error_rate, t_lo, t_hi, overwrite=True)
        if c_hi > n_mine:  # padding tiles of the last ranks' slices
            acc[(rank * per + max(c_lo, n_mine)) * b2:(rank * per + c_hi) * b2].zero_()
        if not exchange:
            continue
        outs = [acc[(r * per + c_lo) * b2:(r * per + c_hi) * b2] for r in range(world)]
        if direct:
            done = torch.cuda.Event()
            done.record(main)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(done)
                dist.all_gather(outs, outs[rank].clone(), group=group)
        else:
            parts = [torch.empty(outs[rank].shape, dtype=acc.dtype) for _ in range(world)]
            dist.all_gather(parts, outs[rank].cpu(), group=group)
            for o, part in zip(outs, parts):
                o.copy_(part)
    if direct:
        main.wait_stream(comm_stream)
        acc.record_stream(comm_stream)
    return acc


def chromosome_cuts(chr_locus_off, locus_entry_off, world):
    """world + 1 chromosome indices: rank r takes chromosomes [cuts[r], cuts[r + 1]). Contiguous, each
    cut at the chromosome boundary nearest to an equal share of the entries (a rank may get none)."""
    import numpy as np
    chr_off = np.asarray(chr_locus_off, dtype=np.int64)
    ent = np.asarray(locus_entry_off, dtype=np.int64)[chr_off]  # entries before each chromosome boundary
    total = int(ent[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.argmin(np.abs(ent - target)))
        cuts.append(max(c, cuts[-1]))
    cuts.append(len(chr_off) - 1)
    return cuts


def chromosome_shard(p, rank, world):
    """The chromosomes of `rank` as a FlatPileup of their own (views of the entry arrays, offsets
    rebased). Position-sorted pileups keep their order; read ids need no change (they are scoped to a
    chromosome already)."""
    import numpy as np
    from .pileup import FlatPileup
    cuts = chromosome_cuts(p.chr_locus_off, p.locus_entry_off, world)
    c0, c1 = cuts[rank], cuts[rank + 1]
    l0, l1 = int(p.chr_locus_off[c0]), int(p.chr_locus_off[c1])
    e0, e1 = int(p.locus_entry_off[l0]), int(p.locus_entry_off[l1])
    return FlatPileup((p.chr_locus_off[c0:c1 + 1].astype(np.int64) - l0).astype(np.uint32), p.locus_pos[l0:l1],
                      (p.locus_entry_off[l0:l1 + 1] - np.uint64(e0)).astype(np.uint64), p.read_ids[e0:e1],
                      p.id_base[e0:e1])


def agree_on_shard_geometry(plan, prepare, world, device="cpu", group=None):
    """Chromosome shards are summed into ONE accumulator, so the ranks must use the same tile edge, the same
    table and the same fixed-point scale, and all three are chosen from a shard's own statistics: the tile
    edge becomes the smallest any rank chose (`prepare(block_cells)` packs again when it differs); the pair
    bound that decides the scale becomes the EXACT bound of the whole pileup -- the per-row sums of squared
    entry counts (plan.cell_squares()) add up over the shards, and the bound is the maximum of the summed
    vector, so N ranks quantise exactly as one GPU does --; the longest read (which table entries are
    reachable) becomes the maximum over the shards. A rank whose shard is empty takes part with zeros.
    `device`: where the exchange lives ("cuda" under RCCL, "cpu" under gloo).
    Returns (block_cells, pair_bound)."""
    import torch
    import torch.distributed as dist
    if not _exchanges(world):
        return plan.block_cells, plan.pair_bound
    b = torch.tensor([plan.block_cells, -plan.max_read_entries], dtype=torch.int64, device=device)
    dist.all_reduce(b, op=dist.ReduceOp.MIN, group=group)
    block_cells, longest = int(b[0].item()), -int(b[1].item())
    if block_cells != plan.block_cells:
        prepare(block_cells)
    sq = plan.cell_squares().to(device)
    dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=group)
    bound = int(sq.max().item()) if sq.numel() else 0
    plan.set_scale_bounds(max(bound, 1), max(longest, 1))
    return block_cells, bound


def chromosome_sharded_accumulate(plan, acc, mutation_rate, homozygous_rate, seq_error_rate, world, group=None,
                                  verify_scale=False):
    """`plan` holds this rank's chromosomes (chromosome_shard; the same block_cells and pair bound on every
    rank: agree_on_shard_geometry): compute all tiles over them into `acc` (overwritten), then sum the
    accumulators of all ranks. A rank with an empty shard still calls accumulate (it sets up the table and
    the scale that finalize needs). Raises when the shared bounds are not in force on this rank (never set, or
    dropped by a prepare from other arrays); verify_scale=True also all-reduces MIN / MAX of the scale the ranks
    used and raises on a mismatch (one small collective and a host synchronisation per call: tests, debugging)."""
    import torch.distributed as dist

    n = plan.acc_elems
    if world > 1 and plan.scale_bounds_state != 1:
        #  accumulators that are added must be quantised at ONE scale: without the shared bounds in
        # force this rank would sum at its shard's own scale -- silently wrong by a power of two when the union
        # crosses a threshold the shard does not
        raise RuntimeError(
            "chromosome_sharded_accumulate: the shared scale bounds are not in force on this rank (%s); call "
            "agree_on_shard_geometry after every prepare from new arrays" % (
                "they were set and a later prepare from other arrays dropped them"
                if plan.scale_bounds_state == 2 else "they were never set"))
    plan.accumulate(acc, mutation_rate, homozygous_rate, seq_error_rate, overwrite=True)
    if verify_scale and _exchanges(world):
        import torch
        dev = acc.device if (acc.is_cuda and dist.get_backend(group) == "nccl") else "cpu"
        chk = torch.tensor([plan.scale_log2, -plan.scale_log2], dtype=torch.int64, device=dev)
        dist.all_reduce(chk, op=dist.ReduceOp.MAX, group=group)
        if int(chk[0].item()) != -int(chk[1].item()):
            raise RuntimeError("chromosome_sharded_accumulate: ranks quantised at different scales (2^%d .. 2^%d)"
                               % (-int(chk[1].item()), int(chk[0].item())))
    if _exchanges(world):
        if acc.is_cuda and dist.get_backend(group) != "nccl":
            summed = acc[:n].cpu()  # rehearsal backends (gloo) move host memory
            dist.all_reduce(summed, op=dist.ReduceOp.SUM, group=group)
            acc[:n].copy_(summed)
        else:
            dist.all_reduce(acc[:n], op=dist.ReduceOp.SUM, group=group)
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
    normalises its rows (cut on cell-block boundaries). `acc` from plan.new_acc(); the listed tiles are overwritten.
    Returns (rows tensor, row_begin)."""
    import torch
    import torch.distributed as dist
    lo, hi = row_range(plan.num_cells, rank, world, plan.block_cells)
    ids = plan.tiles_of_rows(lo, hi)
    plan.accumulate_list(acc, mutation_rate, homozygous_rate, seq_error_rate, ids, overwrite=True)
    local_max = plan.max_of_tiles(acc, ids)
    if _exchanges(world):
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
