"""Two ranks on the one GPU of the test box (gloo rendezvous, both on cuda:0): the sharded path -- tile
ranges per rank, gather of the int64 accumulator, local normalisation -- must reproduce the
single-process matrix bit for bit. RCCL itself needs one GPU per rank: its code paths run here at world size 1
(test_rccl_world_size_1_runs_every_collective_branch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import secedo_amd
        from secedo_amd import distributed as sd
        from tests.pileup_gen import random_pileup

        n = 400
        p = random_pileup(601, n, 2, 500, 60, 1500, dup_frac=0.02)
        torch.cuda.set_device(0)
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.prepare(p, n, 1000, None, 8, block_cells=64)  # 7 blocks -> 28 tiles
            acc = plan.new_acc(pad_tiles_to=world)
            acc.fill_(-123)  # garbage that the sharded path must overwrite
            sd.sharded_accumulate(plan, acc, 0.01, 0.5, 0.01, rank, world)
            out = plan.finalize(acc, "ADD_MIN").cpu().numpy()
            np.save(os.path.join(out_dir, "rank%d.npy" % rank), out)
            if rank == 0:
                full = plan.new_acc()
                plan.accumulate(full, 0.01, 0.5, 0.01)
                np.save(os.path.join(out_dir, "single.npy"), plan.finalize(full, "ADD_MIN").cpu().numpy())
    finally:
        dist.destroy_process_group()


def _spectral_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import secedo_amd
        from secedo_amd import distributed as sd
        from secedo_amd.synth import synth_config

        n = 64
        p = synth_config("C1")  # two clones of 32 cells
        torch.cuda.set_device(0)
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.prepare(p, n, 1000, None, 8)
            acc = plan.new_acc(pad_tiles_to=world)
            sd.sharded_accumulate(plan, acc, 0.01, 0.5, 0.01, rank, world)
            lo, hi = sd.row_range(n, rank, world)  # 3 ranks: 22 + 21 + 21 rows
            rows = plan.finalize_rows(acc, lo, hi, "ADD_MIN")
            # the same row block with nothing gathered: this rank accumulates every tile touching its rows
            own, own_lo = sd.sharded_rows(plan, plan.new_acc(), 0.01, 0.5, 0.01, rank, world, "ADD_MIN")
            own_hi = own_lo + own.shape[0]  # cut on cell-block boundaries: one 64-cell block here, rank 0 has it
            assert (own_lo, own_hi) == sd.row_range(n, rank, world, plan.block_cells)
            assert torch.equal(own, plan.finalize_rows(acc, own_lo, own_hi, "ADD_MIN"))
            vals, vecs, info = sd.sharded_eigenpairs(rows, lo, n, 20, 7)
            assert info["converged"]
            np.save(os.path.join(out_dir, "vals%d.npy" % rank), vals)
            np.save(os.path.join(out_dir, "vecs%d.npy" % rank), vecs.cpu().numpy())
            np.save(os.path.join(out_dir, "rows%d.npy" % rank), rows.cpu().numpy())
            if rank == 0:
                full = plan.finalize(acc, "ADD_MIN")
                np.save(os.path.join(out_dir, "full.npy"), full.cpu().numpy())
                v1, w1, _ = secedo_amd.smallest_eigenpairs(full, 20, 7)
                np.save(os.path.join(out_dir, "vals_single.npy"), v1)
                np.save(os.path.join(out_dir, "vecs_single.npy"), w1.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_matrix_sharded_by_rows_feeds_the_distributed_spectral_step(tmp_path):
    """BASELINE config 5 in small: three ranks (on the one GPU of the box, gloo rendezvous) keep a row
    block each of the normalised matrix (finalize_rows), the spectral step multiplies per rank and
    all-reduces the n x 32 partial products. Every rank must return the same eigenpairs, they must
    agree with the single-process solve on the gathered matrix, and the row blocks must tile it."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    world = 3
    procs = [ctx.Process(target=_spectral_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    full = np.load(tmp_path / "full.npy")
    assert np.array_equal(np.concatenate([np.load(tmp_path / ("rows%d.npy" % r)) for r in range(world)]), full)
    v_single, w_single = np.load(tmp_path / "vals_single.npy"), np.load(tmp_path / "vecs_single.npy")
    for r in range(world):
        vals, vecs = np.load(tmp_path / ("vals%d.npy" % r)), np.load(tmp_path / ("vecs%d.npy" % r))
        assert np.array_equal(vals, np.load(tmp_path / "vals0.npy")) and np.array_equal(vecs, np.load(tmp_path / "vecs0.npy"))
        assert np.max(np.abs(vals - v_single)) <= 1e-9
        # the Fiedler vector splits the two clones on every rank
        side = vecs[:, 1] >= 0
        assert np.all(side[:32] == side[0]) and np.all(side[32:] == side[32]) and side[0] != side[32]
    lap = np.eye(64) - full / np.sqrt(np.outer(full.sum(1), full.sum(1)))
    vals, vecs = np.load(tmp_path / "vals1.npy"), np.load(tmp_path / "vecs1.npy")
    assert np.max(np.abs(lap @ vecs - vecs * vals[:7])) <= 2e-8


def test_two_ranks_reproduce_single_process_bitwise(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    single = np.load(tmp_path / "single.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("rank%d.npy" % r)), single)
    assert np.any(single != 0)


def _chromosome_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import secedo_amd
        from secedo_amd import distributed as sd
        from tests.pileup_gen import random_pileup

        n = 300
        p = random_pileup(611, n, 7, 150, 50, 800, dup_frac=0.03, triple_frac=0.2)  # 7 chromosomes
        torch.cuda.set_device(0)
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            shard = sd.chromosome_shard(p, rank, world)
            plan.prepare(shard, n, 400, None, 2, block_cells=64)  # reads longer than mfl: flushes cut them
            sd.agree_on_shard_geometry(plan, lambda b: plan.prepare(shard, n, 400, None, 2, block_cells=b), world)
            acc = plan.new_acc()
            acc.fill_(-5)  # garbage that the sharded path must overwrite
            sd.chromosome_sharded_accumulate(plan, acc, 0.01, 0.5, 0.01, world)
            np.save(os.path.join(out_dir, "acc%d.npy" % rank), acc.cpu().numpy())
            np.save(os.path.join(out_dir, "rank%d.npy" % rank), plan.finalize(acc, "ADD_MIN").cpu().numpy())
            if rank == 0:
                plan.prepare(p, n, 400, None, 2, block_cells=64)
                full = plan.new_acc()
                plan.accumulate(full, 0.01, 0.5, 0.01)
                np.save(os.path.join(out_dir, "acc_single.npy"), full.cpu().numpy())
                np.save(os.path.join(out_dir, "single.npy"), plan.finalize(full, "ADD_MIN").cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_chromosome_shards_reproduce_single_process_bitwise(tmp_path):
    """Every rank packs and accumulates only its chromosomes (all tiles), the int64 accumulators are summed
    by an all-reduce: the accumulator and the matrix equal the single-process ones bit for bit."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    world = 3
    procs = [ctx.Process(target=_chromosome_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    acc_single, single = np.load(tmp_path / "acc_single.npy"), np.load(tmp_path / "single.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("acc%d.npy" % r)), acc_single)
        assert np.array_equal(np.load(tmp_path / ("rank%d.npy" % r)), single)
    assert np.any(single != 0)


def _deep_and_shallow(n, deep_cells, reads_per_locus=100, loci=1200, seed=99):
    """One deep chromosome per entry of deep_cells (that cell has `reads_per_locus` single-locus reads at each of
    `loci` loci: per-row square sum reads^2 x loci), then one shallow chromosome."""
    from tests.pileup_gen import from_rows
    rng = np.random.default_rng(seed)
    rid = iter(range(1, 10 ** 7))
    chroms = []
    for cell in deep_cells:
        rows = []
        others = np.array([c for c in range(n) if c != cell])
        for l in range(loci):
            ents = [(next(rid), cell, int(rng.integers(0, 4))) for _ in range(reads_per_locus)]
            ents += [(next(rid), int(c), int(rng.integers(0, 4))) for c in rng.choice(others, 6, replace=False)]
            rows.append((1000 + 2000 * l, ents))
        chroms.append(rows)
    chroms.append([(1000 + 2000 * l, [(next(rid), int(c), int(rng.integers(0, 4)))
                                      for c in rng.choice(n, 8, replace=False)]) for l in range(300)])
    return from_rows(chroms)


def test_shards_across_a_scale_threshold_share_one_scale():
    """the fixed-point scale drops below 44 once 1.5 x pair bound x the largest
    per-locus |D| reaches 2^18 (single-locus reads at the default rates: a bound of 1.9e7), and shards that are
    summed must use ONE scale -- that of the whole pileup. Two deep chromosomes of 1.2e7 each:
    (a) both on cell 0: the union (2.4e7) is at scale 43, every shard on its own at 44 -- with the summed
        per-row squares set on every shard (what agree_on_shard_geometry does over RCCL) the shards' accumulators
        add up to the single-process accumulator bit for bit; left to their own scales they do not;
    (b) on cells 0 and 5: the union is at 1.2e7, scale 44 like every shard -- the SUM OF THE SHARDS' MAXIMA
        (2.4e7, what round 2 exchanged) would have lowered the scale to 43 and lost the bitwise match."""
    import secedo_amd
    from secedo_amd import distributed as sd

    n, world = 20, 3
    for deep_cells, union_scale in (((0, 0), 43), ((0, 5), 44)):
        p = _deep_and_shallow(n, deep_cells)
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.prepare(p, n, 1000, None, 1, block_cells=64)
            full = plan.new_acc()
            plan.accumulate(full, 0.01, 0.5, 0.01)
            assert plan.scale_log2 == union_scale
            assert int(plan.cell_squares().max()) == plan.pair_bound
            want = plan.finalize(full, "ADD_MIN").clone()
            shards = [sd.chromosome_shard(p, r, world) for r in range(world)]
            # cuts follow the entries: a deep chromosome alone, an EMPTY shard, the other deep one with the shallow one
            assert [s.n_chr for s in shards] == [1, 0, 2]
            squares, longest, own_scales, maxima = torch.zeros(n, dtype=torch.int64, device="cuda"), 0, [], []
            for s in shards:
                plan.prepare(s, n, 1000, None, 1, block_cells=64)
                squares += plan.cell_squares()
                maxima.append(plan.pair_bound)
                longest = max(longest, plan.max_read_entries)
                part = plan.new_acc()
                plan.accumulate(part, 0.01, 0.5, 0.01)
                own_scales.append(plan.scale_log2)
            assert own_scales == [44, 44, 44]
            if union_scale == 43:  # same row deep in both shards: the union's bound is the sum of the shards' maxima
                assert int(squares.max()) == sum(maxima)
            else:                  # different rows: the sum of the maxima (round 2's exchange) overstates the union
                assert max(maxima) <= int(squares.max()) < sum(maxima) // 2 + (1 << 20)
            total = torch.zeros_like(full)
            for s in shards:
                plan.prepare(s, n, 1000, None, 1, block_cells=64)
                plan.set_scale_bounds(int(squares.max()), longest)
                part = plan.new_acc()
                plan.accumulate(part, 0.01, 0.5, 0.01)
                assert plan.scale_log2 == union_scale
                total += part
            assert torch.equal(total, full)
            assert torch.equal(plan.finalize(total, "ADD_MIN"), want)  # finalize of the last shard: same scale
            # the bounds belong to the pileup they were set on (packing the same arrays again keeps them) ...
            plan.prepare(shards[2], n, 1000, None, 1, block_cells=64)
            plan.accumulate(plan.new_acc(), 0.01, 0.5, 0.01)
            assert plan.scale_log2 == union_scale
            # ... another pileup gets its own scale back
            plan.prepare(shards[0], n, 1000, None, 1, block_cells=64)
            plan.accumulate(plan.new_acc(), 0.01, 0.5, 0.01)
            assert plan.scale_log2 == 44


def test_more_ranks_than_chromosomes_leave_empty_shards():
    """Three chromosomes over eight ranks: five ranks hold an empty pileup (prepared from HBM like the
    others); the per-rank accumulators still add up to the whole one bit for bit."""
    import secedo_amd
    from secedo_amd import distributed as sd
    from tests.pileup_gen import random_pileup

    n, world = 200, 8
    p = random_pileup(77, n, 3, 200, 40, 900, dup_frac=0.03)
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        plan.prepare(p, n, 1000, None, 8, block_cells=64)
        full = plan.new_acc()
        plan.accumulate(full, 0.01, 0.5, 0.01)
        total = torch.zeros_like(full)
        empty = 0
        for r in range(world):
            shard = sd.chromosome_shard(p, r, world)
            plan.prepare_resident(plan.upload(shard, None, n), n, 1000, 8, 64)
            assert plan.num_tiles * 64 * 64 == full.numel()
            part = plan.new_acc()
            if plan.num_entries:
                plan.accumulate(part, 0.01, 0.5, 0.01)
            else:
                empty += 1
            total += part
        torch.cuda.synchronize()
        assert empty == 5
        assert torch.equal(total, full)


def _rccl_world1_worker(port, out_dir):
    """ONE rank, backend nccl (= RCCL) on cuda:0, SECEDO_DIST_EXCHANGE_ALWAYS=1: every collective branch of
    secedo_amd.distributed runs through RCCL exactly as it does with N ranks (in-place all-gather, the chunked
    all-gather into list views on a side stream with record_stream, the geometry all-reduces on the device, the
    all-reduce of the accumulator, the scalar max, the spectral step's partial-product all-reduce)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SECEDO_DIST_EXCHANGE_ALWAYS"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import secedo_amd
        from secedo_amd import distributed as sd
        from secedo_amd.synth import synth_config
        from tests.pileup_gen import random_pileup

        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and sd._exchanges(1)
        n = 400
        p = random_pileup(601, n, 3, 500, 60, 1500, dup_frac=0.02)
        rates = (0.01, 0.5, 0.01)
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.prepare(p, n, 1000, None, 8, block_cells=64)  # 7 blocks -> 28 tiles
            full = plan.new_acc()
            plan.accumulate(full, *rates)
            want = plan.finalize(full, "ADD_MIN").clone()
            # tiles + in-place all_gather_into_tensor
            acc = plan.new_acc(pad_tiles_to=1)
            acc.fill_(-123)
            sd.sharded_accumulate(plan, acc, *rates, 0, 1)
            assert torch.equal(acc, full)
            # tiles, the exchange in chunks on a second stream (all_gather into a list of views)
            for chunks in (2, 3, 5):
                acc.fill_(-7)
                sd.sharded_accumulate_overlapped(plan, acc, *rates, 0, 1, chunks=chunks)
                torch.cuda.synchronize()
                assert torch.equal(acc, full), chunks
            assert torch.equal(plan.finalize(acc, "ADD_MIN"), want)
            # chromosomes: geometry agreed over RCCL (device tensors), all-reduce of the accumulator
            shard = sd.chromosome_shard(p, 0, 1)
            plan.prepare(shard, n, 1000, None, 8, block_cells=64)
            b, bound = sd.agree_on_shard_geometry(
                plan, lambda bc: plan.prepare(shard, n, 1000, None, 8, block_cells=bc), 1, "cuda")
            assert (b, bound) == (64, plan.pair_bound) and plan.scale_bounds_state == 1
            acc.fill_(-5)
            sd.chromosome_sharded_accumulate(plan, acc, *rates, 1, verify_scale=True)
            assert torch.equal(acc, full)
            # rows kept sharded (scalar max all-reduce on the device) + the distributed spectral step
            rows, lo = sd.sharded_rows(plan, plan.new_acc(), *rates, 0, 1, "ADD_MIN")
            assert lo == 0 and torch.equal(rows, want)
        p1 = synth_config("C1")
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.prepare(p1, 64, 1000, None, 8)
            rows, lo = sd.sharded_rows(plan, plan.new_acc(), *rates, 0, 1, "ADD_MIN")
            vals, vecs, info = sd.sharded_eigenpairs(rows, lo, 64, 20, 7)
            v1, w1, _ = secedo_amd.smallest_eigenpairs(rows, 20, 7)
            assert info["converged"] and np.max(np.abs(vals - v1)) <= 1e-9
        with open(os.path.join(out_dir, "ok"), "w") as fh:
            fh.write("rccl world %d backend %s\n" % (dist.get_world_size(), dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_1_runs_every_collective_branch(tmp_path):
    """no line of the RCCL code had ever executed. A world-size-1 RCCL group on the box's one GPU
    (a child process, so that the test process never initialises RCCL) drives sharded_accumulate,
    sharded_accumulate_overlapped, agree_on_shard_geometry(device="cuda") + chromosome_sharded_accumulate,
    sharded_rows and sharded_eigenpairs through their `direct` (nccl) branches, each bitwise equal to the
    single-process result."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    proc = ctx.Process(target=_rccl_world1_worker, args=(port, str(tmp_path)))
    proc.start()
    proc.join(600)
    assert proc.exitcode == 0
    assert (tmp_path / "ok").read_text().startswith("rccl world 1 backend nccl")


def test_bounds_dropped_by_a_prepare_from_copied_arrays_are_refused():
    """the shared scale bounds belong to the pileup held when they were set; a prepare from a
    re-created (equal) array drops them, and setting them before the pileup is set loses them too. Neither may pass
    silently: scale_bounds_state says so and chromosome_sharded_accumulate refuses to add such an accumulator."""
    import copy

    import secedo_amd
    from secedo_amd import distributed as sd
    from tests.pileup_gen import random_pileup

    n = 120
    p = random_pileup(613, n, 3, 150, 30, 800)
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        assert plan.scale_bounds_state == 0
        plan.prepare(p, n, 1000, None, 4, block_cells=64)
        plan.set_scale_bounds(1 << 30, 4)
        assert plan.scale_bounds_state == 1
        plan.prepare(p, n, 1000, None, 4, block_cells=64)       # the same arrays again: kept
        assert plan.scale_bounds_state == 1
        acc = plan.new_acc()
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        shared_scale = plan.scale_log2
        assert shared_scale < 44
        q = copy.deepcopy(p)                                    # equal data, other arrays
        plan.prepare(q, n, 1000, None, 4, block_cells=64)
        assert plan.scale_bounds_state == 2
        with pytest.raises(RuntimeError, match="scale bounds"):
            sd.chromosome_sharded_accumulate(plan, acc, 0.01, 0.5, 0.01, world=2)
        plan.set_scale_bounds(1 << 30, 4)                       # set again after the prepare: in force
        assert plan.scale_bounds_state == 1
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        assert plan.scale_log2 == shared_scale
        plan.set_scale_bounds(0, 0)
        assert plan.scale_bounds_state == 0
    with secedo_amd.SimilarityMatrixPlan(0) as plan:            # bounds before the pileup: taken away by set_pileup
        plan.set_scale_bounds(1 << 30, 4)
        plan.prepare(p, n, 1000, None, 4, block_cells=64)
        assert plan.scale_bounds_state == 2
