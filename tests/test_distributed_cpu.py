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

    def accumulate(self, acc, eps, h, theta, lo, hi):
        b2 = self.block_cells ** 2
        for t in range(lo, hi):
            acc[t * b2:(t + 1) * b2] += t + 1


def _worker(rank, world, port, num_tiles, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = FakePlan(num_tiles, 4)
        per = sd.tiles_per_rank(num_tiles, world)
        acc = torch.full((per * world * 16,), -7, dtype=torch.int64)  # garbage that must disappear
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


@pytest.mark.parametrize("num_tiles", [5, 36])
def test_sharded_accumulate_gloo_world2(num_tiles):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_tiles, q)) for r in range(2)]
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
