"""The N > 1 path on CPU: tile partition + the all-gather composition with gloo, world_size 2.

No GPU here, so the per-rank accumulate is replaced by a stand-in that fills the rank's tiles with
known integers; what is under test is secedo_amd.distributed (ranges, padding, in-place gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from secedo_amd import distributed as sd


def test_tile_ranges_cover_exactly_once():
    for nt in (1, 2, 7, 36, 136, 2016):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(nt, dtype=np.int32)
            for r in range(world):
                lo, hi = sd.tile_range(nt, r, world)
                assert 0 <= lo <= hi <= nt
                seen[lo:hi] += 1
            assert np.all(seen == 1)
            assert sd.tiles_per_rank(nt, world) * world >= nt


class FakePlan:
    """Duck-typed SimilarityMatrixPlan: accumulate() writes tile_index + 1 into every tile it owns."""

    def __init__(self, num_tiles, block_cells):
        self.num_tiles, self.block_cells = num_tiles, block_cells

    def accumulate(self, acc, eps, h, theta, lo, hi, overwrite=False):
        b2 = self.block_cells ** 2
        for t in range(lo, hi):
            if overwrite:
                acc[t * b2:(t + 1) * b2] = t + 1
            else:
                acc[t * b2:(t + 1) * b2] += t + 1


def _worker(rank, world, port, num_tiles, chunks, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = FakePlan(num_tiles, 4)
        per = sd.tiles_per_rank(num_tiles, world)
        acc = torch.full((per * world * 16,), -7, dtype=torch.int64)  # garbage that must disappear
        if chunks:
            sd.sharded_accumulate_overlapped(plan, acc, 0.01, 0.5, 0.01, rank, world, chunks=chunks)
        else:
            sd.sharded_accumulate(plan, acc, 0.01, 0.5, 0.01, rank, world)
        expect = torch.zeros_like(acc)
        for t in range(num_tiles):
            expect[t * 16:(t + 1) * 16] = t + 1
        ok = torch.equal(acc, expect)
        flags = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(flags, torch.tensor([int(ok)], dtype=torch.int64))
        if rank == 0:
            out.put([int(f.item()) for f in flags])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_tiles,chunks", [(5, 0), (36, 0), (5, 2), (36, 4), (37, 3), (2, 4)])
def test_sharded_accumulate_gloo_world2(num_tiles, chunks):
    """chunks = 0: one all-gather after the rank's tiles; otherwise the chunked exchange (on RCCL it overlaps the
    accumulation of the next chunk): both must leave every tile at its place, bit for bit, padding zeroed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_tiles, chunks, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) == [1, 1]


def test_row_ranges_partition_the_matrix():
    """Row blocks of the matrix kept sharded (BASELINE config 5): contiguous, disjoint, covering, even."""
    from secedo_amd.distributed import row_range
    for n in (1, 7, 64, 1000, 32000):
        for world in (1, 2, 3, 8):
            edges = [row_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
            aligned = [row_range(n, r, world, 128) for r in range(world)]
            assert aligned[0][0] == 0 and aligned[-1][1] == n
            assert all(aligned[r][1] == aligned[r + 1][0] for r in range(world - 1))
            assert all(lo % 128 == 0 or lo == n for lo, _ in aligned)


def test_chromosome_shards_sum_to_the_whole_matrix():
    """The chromosome split is exact: the raw (pre-normalisation) matrices of the shards, computed by the
    oracle -- which carries the reference's `completed` counter from chromosome to chromosome -- add up to
    the raw matrix of the whole pileup (up to the rounding of the oracle's double sums, 1e-13; the device
    path accumulates integers and is compared bit for bit in tests/test_gpu_distributed.py); shards are
    contiguous, disjoint and cover everything."""
    from oracle import bindings as ob
    from tests.pileup_gen import random_pileup
    n = 40
    for seed, n_chr, mfl, threads in ((5, 5, 1000, 1), (6, 7, 150, 2), (7, 3, 1000, 8)):
        p = random_pileup(seed, n, n_chr, 120, 12, 200, dup_frac=0.05, triple_frac=0.2)
        _, raw = ob.oracle_compute(p, n, mfl, None, 0.01, 0.5, 0.02, threads, "ADD_MIN", want_raw=True)
        for world in (2, 3, 4, 9):
            cuts = sd.chromosome_cuts(p.chr_locus_off, p.locus_entry_off, world)
            assert cuts[0] == 0 and cuts[-1] == p.n_chr and all(a <= b for a, b in zip(cuts, cuts[1:]))
            total = np.zeros_like(raw)
            entries = 0
            for r in range(world):
                shard = sd.chromosome_shard(p, r, world)
                entries += shard.n_entries
                assert shard.n_chr == cuts[r + 1] - cuts[r]
                if shard.n_entries:
                    _, part = ob.oracle_compute(shard, n, mfl, None, 0.01, 0.5, 0.02, threads, "ADD_MIN",
                                                want_raw=True)
                    total += part
            assert entries == p.n_entries
            assert np.max(np.abs(total - raw)) <= 1e-11 * max(1.0, np.max(np.abs(raw)))


class FakeShardPlan:
    """accumulate() gives rank + 1 in every element: the all-reduce must leave 1 + 2 + ... + world."""

    def __init__(self, rank, n, scale_bounds_state=1, scale_log2=44):
        self.rank, self.acc_elems, self.num_entries = rank, n, 1
        self.scale_bounds_state, self.scale_log2 = scale_bounds_state, scale_log2

    def accumulate(self, acc, eps, h, theta, overwrite=False):
        if overwrite:
            acc[:self.acc_elems] = self.rank + 1
        else:
            acc[:self.acc_elems] += self.rank + 1


def _shard_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        acc = torch.full((48,), -3, dtype=torch.int64)  # garbage that must disappear; 8 elements of padding
        sd.chromosome_sharded_accumulate(FakeShardPlan(rank, 40), acc, 0.01, 0.5, 0.01, world, verify_scale=True)
        ok = bool(torch.all(acc[:40] == world * (world + 1) // 2))
        #  a rank whose shared bounds are not in force -- never set, or dropped by a prepare from a
        # re-created array -- must not add its accumulator to the others': refused locally, before any collective
        for state in (0, 2):
            try:
                sd.chromosome_sharded_accumulate(FakeShardPlan(rank, 40, state), acc.clone(), 0.01, 0.5, 0.01, world)
                ok = False
            except RuntimeError as e:
                ok = ok and "scale bounds" in str(e)
        # ... and ranks that DID quantise at different scales are found out by verify_scale
        try:
            sd.chromosome_sharded_accumulate(FakeShardPlan(rank, 40, 1, 44 - rank), acc.clone(), 0.01, 0.5, 0.01, world,
                                             verify_scale=True)
            ok = False
        except RuntimeError as e:
            ok = ok and "different scales" in str(e)
        flags = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(flags, torch.tensor([int(ok)], dtype=torch.int64))
        if rank == 0:
            out.put([int(f.item()) for f in flags])
    finally:
        dist.destroy_process_group()


class FakeGeometryPlan:
    """What agree_on_shard_geometry needs of a plan: a tile edge, the per-row squares and the longest read of
    the shard."""

    def __init__(self, block_cells, squares, max_read_entries):
        self.block_cells, self.squares, self.max_read_entries = block_cells, squares, max_read_entries
        self.bounds_set, self.repacked = None, []

    @property
    def pair_bound(self):
        return max(self.squares)

    def cell_squares(self):
        return torch.tensor(self.squares, dtype=torch.int64)

    def set_scale_bounds(self, bound, longest):
        self.bounds_set = (bound, longest)


def _geometry_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0: a deep row 0, reads of up to 3 entries; rank 1: a deep row 1, reads of 9 (an empty shard has
        # all zeros and is covered on the GPU: tests/test_gpu_distributed.py)
        plan = FakeGeometryPlan([128, 64][rank], [[300000, 10, 0], [0, 200000, 7]][rank], [3, 9][rank])

        def prepare(b):
            plan.repacked.append(b)
            plan.block_cells = b

        got = sd.agree_on_shard_geometry(plan, prepare, world)
        res = torch.tensor([got[0], got[1], plan.bounds_set[0], plan.bounds_set[1], len(plan.repacked)], dtype=torch.int64)
        parts = [torch.zeros_like(res) for _ in range(world)]
        dist.all_gather(parts, res)
        if rank == 0:
            out.put([p.tolist() for p in parts])
    finally:
        dist.destroy_process_group()


def test_shards_agree_on_tile_edge_and_fixed_point_scale_gloo_world2():
    """Chromosome shards are added into one accumulator: every rank must quantise with the same scale -- the
    EXACT pair bound of the union decides it: the per-row squares add up over the shards and the bound is the
    maximum of the sum (300010, not the 500000 the sum of the shards' maxima would give: ADVICE r02) --, reach the
    same table entries (the longest read of any shard) and use the same tile edge (the smallest)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_geometry_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # (tile edge, union bound, bound and longest read handed to the plan, re-packs): rank 0 re-packed with 64
    assert q.get(timeout=10) == [[64, 300000, 300000, 9, 1], [64, 300000, 300000, 9, 0]]


def test_chromosome_sharded_accumulate_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) == [1, 1]
