"""Two ranks on the one GPU of the test box (gloo rendezvous, both on cuda:0): the sharded path -- tile
ranges per rank, gather of the int64 accumulator, local normalisation -- must reproduce the
single-process matrix bit for bit. (RCCL itself needs one GPU per rank; the driver exercises it.)"""
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
