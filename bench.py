#!/usr/bin/env python3
"""bench.py -- throughput of the similarity-matrix hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C1] [--clustered]

One step = one pass of the hot path over the synthetic pileup already resident in HBM:
zero the accumulator, accumulate every tile (this rank's tile range when N > 1), all-gather the
accumulator over RCCL (N > 1), normalise + mirror into the dense N x N fp64 matrix.
The metric is (read pair, shared locus) updates per second (BASELINE.json "cell-pair x locus
updates/sec"; one update = one x_s++/x_d++ of reference similarity_matrix.cpp:225).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(p, n_cells, mfl, rates, threads, budget_updates):
    """Reference (oracle/_ref) or the C restatement, timed on this host on a bounded sample."""
    import numpy as np
    from oracle import bindings as ob
    from secedo_amd.pileup import FlatPileup

    # bounded sample: a prefix of the loci whose pair-locus count stays below the budget
    cov = np.diff(p.locus_entry_off.astype(np.int64))
    cum = np.cumsum(cov * (cov - 1) // 2)
    n_loci = int(np.searchsorted(cum, budget_updates, side="right"))
    n_loci = max(1, min(n_loci, p.n_loci))
    if n_loci < p.n_loci:
        chr_off = np.minimum(p.chr_locus_off, n_loci).astype(np.uint32)
        e = int(p.locus_entry_off[n_loci])
        sample = FlatPileup(chr_off, p.locus_pos[:n_loci], p.locus_entry_off[:n_loci + 1],
                            p.read_ids[:e], p.id_base[:e])
        what = "first %d of %d loci" % (n_loci, p.n_loci)
    else:
        sample, what = p, "full workload"
    eps, h, theta = rates
    # updates of the sample, from the oracle's exact counter
    ob.oracle_compute(sample, n_cells, mfl, None, eps, h, theta, threads, "ADD_MIN")
    updates = ob.oracle_last_updates()
    if ob.have_ref():
        kind = "reference"
        best = None
        for t in (1, threads):  # the reference scales negatively with threads: report the better
            t0 = time.perf_counter()
            ob.ref_compute(sample, n_cells, mfl, None, eps, h, theta, t, "ADD_MIN")
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, t)
        dt, cores = best
    else:
        kind, cores = "port", 1
        t0 = time.perf_counter()
        ob.oracle_compute(sample, n_cells, mfl, None, eps, h, theta, threads, "ADD_MIN")
        dt = time.perf_counter() - t0
    return {"value": updates / dt, "unit": "updates/s", "cores": cores, "kind": kind,
            "sample": "%s (%d updates, %.1f s)" % (what, updates, dt),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C5"])
    ap.add_argument("--clustered", action="store_true", help="gap_max=300 variant (~2.8 loci/read)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to rehearse the N > 1 "
                         "flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--tiles", action="store_true",
                    help="N > 1: deal out the output tiles (replicated packing + all-gather) even when the pileup "
                         "has chromosomes enough to be split by chromosomes (the default then)")
    ap.add_argument("--packed-resident", action="store_true",
                    help="keep the PACKED pileup resident and leave the packing out of the step "
                         "(steady state of repeated accumulations; default: the step starts from the raw "
                         "flat pileup in HBM and includes the device-side packing)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import secedo_amd
    from secedo_amd import distributed as sd
    from secedo_amd.synth import CONFIGS, synth_config

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    n_cells, n_loci, n_chr, gap, prob = CONFIGS[args.workload]
    mfl, threads, rates, norm = 1000, 8, (0.01, 0.5, 0.01), "ADD_MIN"
    p = synth_config(args.workload, clustered=args.clustered)

    # N > 1, strong scaling of ONE matrix. With chromosomes enough the pileup is split by chromosomes (reads,
    # flushes and the tail rule never cross one): a rank packs and accumulates only its chromosomes, for all
    # tiles, and the int64 accumulators are summed by one all-reduce -- the packing is divided too.
    # Otherwise the tiles are dealt out, every rank packs the whole pileup, one all-gather.
    by_chromosome = world > 1 and p.n_chr >= world and not args.tiles
    mine = sd.chromosome_shard(p, rank, world) if by_chromosome else p

    plan = secedo_amd.SimilarityMatrixPlan(local_rank)
    resident = plan.upload(mine, None, n_cells)  # the raw flat pileup, in HBM before the clock starts
    t0 = time.perf_counter()
    plan.prepare_resident(resident, n_cells, mfl, threads)
    torch.cuda.synchronize()
    prepare_s = time.perf_counter() - t0
    block_cells = 0  # the library's choice
    if by_chromosome:
        # the ranks must agree on the tile edge (it is chosen from the shard's statistics): the smallest
        b = torch.tensor([plan.block_cells], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(b, op=dist.ReduceOp.MIN)
        block_cells = int(b.item())
        plan.prepare_resident(resident, n_cells, mfl, threads, block_cells)
    acc = plan.new_acc(pad_tiles_to=1 if by_chromosome else world)
    out = torch.empty((n_cells, n_cells), dtype=torch.float64, device="cuda:%d" % local_rank)
    my_tiles = (0, plan.num_tiles) if by_chromosome else sd.tile_range(plan.num_tiles, rank, world)

    def step():
        if not args.packed_resident:
            plan.prepare_resident(resident, n_cells, mfl, threads, block_cells)
        if by_chromosome:
            sd.chromosome_sharded_accumulate(plan, acc, *rates, world)
        else:
            sd.sharded_accumulate(plan, acc, *rates, rank, world)
        plan.finalize(acc, norm, out)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # exact work counters of one pass (integer, identical every step)
    local_updates, local_pairs = plan.last_counts()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    # the accumulate kernel's own duration: HIP events recorded by the library on the launch stream
    # around the last launch of the timed region, plus torch events over a few more launches below
    last_ms = plan.last_accumulate_ms()

    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    cnt = torch.tensor([local_updates, local_pairs, plan.num_entries if by_chromosome or rank == 0 else 0,
                        plan.num_reads if by_chromosome or rank == 0 else 0], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    updates, pairs, kept_entries, reads = (int(v.item()) for v in cnt)

    # phase times of one step, events on the launch stream (torch's current one)
    phase = {}
    def repack():
        plan.prepare_resident(resident, n_cells, mfl, threads, block_cells)

    def refinalize():
        plan.finalize(acc, norm, out)

    for name, fn in (("pack_ms", repack), ("finalize_ms", refinalize)):
        ts = []
        if name == "finalize_ms":
            plan.accumulate(acc, *rates, *my_tiles)
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        phase[name] = sorted(ts)[len(ts) // 2]
    # per-launch kernel time over K launches with events on the launch stream
    lo, hi = my_tiles
    evs = []
    for _ in range(min(args.steps, 10)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        plan.accumulate(acc, *rates, lo, hi)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ev_ms = sorted(a.elapsed_time(b) for a, b in evs)
    kern_ms = ev_ms[len(ev_ms) // 2]

    if rank == 0:
        traffic = None  # HBM bytes per launch from rocprofv3 PMC passes (profiles/r01_traffic.json)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
                t = json.load(fh).get(args.workload if not args.clustered and world == 1 else "")
            if t:
                traffic = (t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024
        except (OSError, ValueError, KeyError):
            traffic = None
        E, L, N = plan.num_entries, plan.num_loci, n_cells
        # this rank's launch: its tiles of the whole pileup, or all tiles of its chromosomes
        b_alg = (16 * local_updates + 6 * E + 4 * L + 16 * N * N if by_chromosome
                 else 16 * local_updates + 6 * E / world + 4 * L + 16 * N * N / world)
        achieved = b_alg / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "cell-pair x locus updates/sec (similarity matrix)",
            "value": updates * args.steps / elapsed,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic (SYNTH-v1, seed 42)",
            "config": {"workload": "%s: %d cells x %d loci%s" % (
                args.workload, n_cells, n_loci, " clustered" if args.clustered else ""),
                "entries": int(p.n_entries), "kept_entries": kept_entries, "reads": reads,
                "updates_per_step": updates, "read_pairs_per_step": pairs,
                "block_cells": plan.block_cells, "tiles": plan.num_tiles,
                "normalization": norm, "max_fragment_length": mfl, "num_threads": threads,
                "parallelism": ("single GPU" if world == 1 else
                                "chromosomes/%d (packing + accumulation) + all-reduce of the int64 accumulator" % world
                                if by_chromosome else "tiles/%d + all-gather" % world),
                # tiles mode: every rank packs the whole pileup, only the pair accumulation is divided
                "replicated_ms_per_step": phase.get("pack_ms") if world > 1 and not by_chromosome else None},
            "wall_s_full_matrix": elapsed / args.steps,
            "dense_equivalent_cell_pair_locus_slots_per_s":
                n_cells * (n_cells - 1) / 2 * n_loci * args.steps / elapsed,
            "step_includes_packing": not args.packed_resident,
            "packing": "device" if plan.used_device_packing else "host",
            "phase_ms": {"pack": phase["pack_ms"], "accumulate": kern_ms, "finalize": phase["finalize_ms"]},
            "first_prepare_s": prepare_s,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "accumulate_tiles", "kernel_ms": kern_ms,
                         "kernel_ms_last_timed_step": last_ms,
                         "algorithmic_bytes": b_alg},
        }
        if not args.no_cpu_baseline:
            # the drop-in call itself: host buffers in, host matrix out (create, H2D, pack, accumulate,
            # normalise, D2H, destroy) -- PCIe-inclusive, never `value`
            # (the library keeps the call's device buffers for the next call: first and repeated call)
            t0 = time.perf_counter()
            secedo_amd.compute_similarity_matrix(p, n_cells, mfl, None, *rates, threads, "", norm)
            line["one_shot_first_call_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            secedo_amd.compute_similarity_matrix(p, n_cells, mfl, None, *rates, threads, "", norm)
            line["one_shot_host_call_s"] = time.perf_counter() - t0
            line["cpu_baseline"] = cpu_baseline(p, n_cells, mfl, rates, threads, 2.5e8)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
