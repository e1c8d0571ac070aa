#!/usr/bin/env python3
"""bench.py -- throughput of the similarity-matrix hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C5|C1] [--clustered]

One step = one pass of the hot path over the synthetic pileup already resident in HBM: device packing
of the raw flat pileup (read assembly, flush schedule, tiles), zero the accumulator, accumulate the
tiles, (N > 1: RCCL exchange), normalise + mirror into the dense N x N fp64 matrix.
The metric is (read pair, shared locus) updates per second (BASELINE.json "cell-pair x locus
updates/sec"; one update = one x_s++/x_d++ of reference similarity_matrix.cpp:225).
The default workload is C3 = BASELINE.json configs[2] (8000 cells x 100K loci), the configuration the
north-star target is stated on; it fits one GPU.

`python bench.py --gpus N` with N > 1 starts its own N ranks (one process per GPU, torch.distributed.run,
rendezvous on 127.0.0.1) when it was not itself started by a launcher; started under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses the ranks it was given.
With N > 1 both ways of sharing the matrix are timed, K steps each:
  * tiles: the output tiles are dealt to the ranks, every rank packs the whole (replicated) pileup, one
    all-gather of the int64 tile-major accumulator (the north-star's block partition + all-gather);
  * chromosomes: every rank packs and accumulates only its chromosomes, for all tiles, one all-reduce
    (sum) of the accumulator.
`value` is the faster of the two (named in config.parallelism); both are in `partitionings`.
`--workload C5 --gpus N` times BASELINE configs[4] as written instead: the matrix kept sharded by rows (no
all-gather) and fed to the distributed spectral step, both phases reported.
The K-step block is timed `--repeats` times (5): the line carries the median block, all blocks and the spread.
After the timed region the last output is held against the compiled reference's sampled entries where a digest
exists (C3, C2): `parity_checked`.
`roofline` is stated against the limit that binds the dominant kernel: LDS atomics at random addresses (peak
measured by tools/lds_atomic_bench on the same box); the HBM figures (measured traffic, and SURVEY 8d's
algorithmic-bytes accounting under `b_alg`) are beside it.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_ROUND = "r04"   # profiles/<round>_*.json: the counter-derived inputs of the roofline record
LIVE_COUNTERS_BUDGET_S = 240.0  # all rocprofv3 --pmc child passes of a default run together


def _pmc_pass(counters, program, out_dir, timeout):
    """One rocprofv3 --pmc pass (counters only, the program directly behind `--`) in a child process; returns
    {kernel name: {counter: per-dispatch average}} from every counter_collection.csv the pass left."""
    import csv
    import glob
    env = dict(os.environ, SECEDO_BENCH_NO_CHILD="1", TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out_dir, "--"] + list(program)
    subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
    acc = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def live_counters(args, kernel_needle, kernel_ms):
    """HBM bytes per launch and the SQ / LDS counters of the dominant kernel, measured in THIS run: rocprofv3 --pmc passes
    over a short child run of this same script (separate passes, counters only -- the HBM recipe of MI355X_MICROARCH.md:
    FETCH_SIZE and WRITE_SIZE each on its own, both calibrated on a known 1 GiB stream in the same run). Returns
    (traffic record, counters record) in the shape of profiles/<round>_traffic.json / _counters.json, or (None, None)
    when no profiler is at hand (the caller falls back to the committed files and says so)."""
    import shutil
    import tempfile
    calib = os.path.join(ROOT, "tools", "fetch_calib.bin")
    if os.environ.get("SECEDO_BENCH_NO_CHILD") or not shutil.which("rocprofv3") or not os.path.exists(calib):
        return None, None
    child = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", "3", "--warmup", "1",
             "--repeats", "1", "--no-cpu-baseline"] + (["--clustered"] if args.clustered else [])
    tmp = tempfile.mkdtemp(prefix="secedo_pmc_", dir="/tmp")
    #  six child runs: ONE budget for all of them (LIVE_COUNTERS_BUDGET_S, about 60 s are used on C3), a
    # pass gets what is left of it -- the default command must stay within minutes even when a pass hangs
    deadline = time.monotonic() + LIVE_COUNTERS_BUDGET_S

    def left(cap):
        remaining = deadline - time.monotonic()
        if remaining < 5:
            raise subprocess.TimeoutExpired("rocprofv3 --pmc passes", LIVE_COUNTERS_BUDGET_S)
        return min(cap, remaining)
    try:
        GIB_KIB = float(1 << 20)
        pick = lambda table, needle: next(v for k, v in table.items() if needle in k)
        cal_f = _pmc_pass(["FETCH_SIZE"], [calib], os.path.join(tmp, "cf"), left(60))
        cal_w = _pmc_pass(["WRITE_SIZE"], [calib], os.path.join(tmp, "cw"), left(60))
        f4 = pick(cal_f, "read4")["FETCH_SIZE"] / GIB_KIB
        w16 = pick(cal_w, "write16")["WRITE_SIZE"] / GIB_KIB
        fetch = pick(_pmc_pass(["FETCH_SIZE"], child, os.path.join(tmp, "f"), left(120)), kernel_needle)["FETCH_SIZE"]
        write = pick(_pmc_pass(["WRITE_SIZE"], child, os.path.join(tmp, "w"), left(120)), kernel_needle)["WRITE_SIZE"]
        traffic = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                   "hbm_bytes_per_launch": (fetch / f4 + write / w16) * 1024.0,
                   "calibration": {"FETCH_SIZE_per_true_KiB_read_4B_per_lane": f4,
                                   "WRITE_SIZE_per_true_KiB_write_16B_per_lane": w16},
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this run, calibrated in this run"}
        counters = {}
        for group in (["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                       "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
                      ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                       "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU"]):
            counters.update(pick(_pmc_pass(group, child, os.path.join(tmp, "s" + group[0]), left(120)), kernel_needle))
        counters["kernel_cycles"] = kernel_ms * 1e-3 * 2.4e9  # this run's HIP-event duration x the shader clock
        counters["source"] = "rocprofv3 --pmc, two passes of this run"
        return traffic, counters
    except (OSError, subprocess.SubprocessError, StopIteration, KeyError, ValueError, ZeroDivisionError):
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def lds_atomic_peak():
    """Peak of the instruction the pair kernel is bound by -- random-address ds_add_u32 -- in 1e9 atomics/s for
    the whole chip: measured live by tools/lds_atomic_bench.bin (a child process, < 1 s; built by make) where it
    runs, else the committed measurement. -> (peak, source, whole record)"""
    exe = os.path.join(ROOT, "tools", "lds_atomic_bench.bin")
    try:
        if os.environ.get("SECEDO_BENCH_NO_CHILD"):  # under a profiler: no second process in the trace
            raise OSError("child processes disabled")
        r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        return rec["peak_random_gatomic_per_s"], "tools/lds_atomic_bench.bin --json, this run", rec
    except (OSError, subprocess.SubprocessError, ValueError, IndexError, KeyError):
        pass
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_lds_atomic_peak.json")) as fh:
            rec = json.load(fh)
        return rec["peak_random_gatomic_per_s"], "profiles/%s_lds_atomic_peak.json" % PROFILE_ROUND, rec
    except (OSError, ValueError, KeyError):
        return None, None, None


def reference_digest(workload, clustered, n_entries, rates, mfl, threads):
    """Sampled entries of the COMPILED REFERENCE's output for this workload (tests/golden/c?_reference_digest.npz,
    written by oracle/gen_golden.py from oracle/_ref in the build container), if the parameters are the digest's."""
    import numpy as np
    if workload not in ("C2", "C3"):
        return None
    name = "%s%s_reference_digest.npz" % (workload.lower(), "_clustered" if clustered else "")
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", name))
    except OSError:
        return None
    want = [mfl, rates[0], rates[1], rates[2], threads, 0]
    if int(z["n_entries"]) != n_entries or [float(v) for v in z["params"][1:]] != [float(v) for v in want]:
        return None
    return z


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step block is timed this many times; the line reports the median block and the spread")
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C5"])
    ap.add_argument("--clustered", action="store_true", help="gap_max=300 variant (~2.8 loci/read)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="take roofline.traffic and the SQ counters from profiles/ instead of measuring them in this run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to rehearse the N > 1 "
                         "flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--group", action="store_true",
                    help="N = 1 only: run the step of the N > 1 job -- process group (world size 1), the three "
                         "partitionings, every collective issued (SECEDO_DIST_EXCHANGE_ALWAYS) -- so that the RCCL code "
                         "paths execute on a one-GPU box; the line's rccl_world_size then comes from a real group")
    ap.add_argument("--only", default="both", choices=["both", "tiles", "chromosomes"],
                    help="N > 1: time only one of the two partitionings")
    ap.add_argument("--chunks", type=int, default=4,
                    help="N > 1, tiles: also time the all-gather issued in this many chunks on a second stream while "
                         "the next chunk accumulates (1 = only the single all-gather at the end)")
    ap.add_argument("--gathered", action="store_true",
                    help="N > 1 with --workload C5: gather the matrix like C3 instead of BASELINE configs[4] "
                         "(sharded rows + distributed spectral step)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: start the ranks, rendezvous (gloo), shard the "
                         "pileup both ways, exchange the shard sizes, print them -- no HIP call anywhere")
    ap.add_argument("--packed-resident", action="store_true",
                    help="keep the PACKED pileup resident and leave the packing out of the step "
                         "(steady state of repeated accumulations; default: the step starts from the raw "
                         "flat pileup in HBM and includes the device-side packing)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """The parent of `python bench.py --gpus N`: it has not touched the GPU (no HIP call, no torch.cuda
    call), starts N fresh rank processes through torch.distributed.run and passes their output on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this host
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world, emit):
    """What the ranks do before any GPU work, on the CPU: rendezvous, shard, exchange. Used by the tests to
    check that `python bench.py --gpus N` starts N ranks that find each other and cover the work once."""
    import torch
    import torch.distributed as dist
    from secedo_amd import distributed as sd
    from secedo_amd.synth import CONFIGS, synth_config

    if world > 1:
        dist.init_process_group("gloo")
    p = synth_config(args.workload, clustered=args.clustered)
    shard = sd.chromosome_shard(p, rank, world) if world > 1 else p
    n_blocks = -(-CONFIGS[args.workload][0] // 128)
    lo, hi = sd.tile_range(n_blocks * (n_blocks + 1) // 2, rank, world)
    mine = torch.tensor([shard.n_entries, shard.n_chr, hi - lo], dtype=torch.int64)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, mine)
    else:
        parts = [mine]
    if rank == 0:
        emit({"dry_run": True, "n_gpus": world, "world_size": world,
              "backend": "gloo" if world > 1 else "none",
              "entries": int(p.n_entries), "chromosomes": int(p.n_chr),
              "tiles": n_blocks * (n_blocks + 1) // 2,
              "shard_entries": [int(t[0]) for t in parts],
              "shard_chromosomes": [int(t[1]) for t in parts],
              "shard_tiles": [int(t[2]) for t in parts]})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    import secedo_amd
    from secedo_amd import distributed as sd
    from secedo_amd.synth import CONFIGS, synth_config

    # ONE JSON line on stdout: libraries that write banners to file descriptor 1 (RCCL prints its version there when
    # the first communicator is made) get stderr instead; the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        os.write(json_fd, (json.dumps(record) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world, emit)
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.group  # a process group exists and the collectives are issued
    if dist_on:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ["SECEDO_DIST_EXCHANGE_ALWAYS"] = "1"
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    n_cells, n_loci, n_chr, gap, prob = CONFIGS[args.workload]
    mfl, threads, rates, norm = 1000, 8, (0.01, 0.5, 0.01), "ADD_MIN"
    p = synth_config(args.workload, clustered=args.clustered)
    out = torch.empty((n_cells, n_cells), dtype=torch.float64, device="cuda:%d" % local_rank)

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def run_mode(by_chromosome, overlapped=False):
        """W warm-up steps, then K timed steps of one partitioning. Returns its record and live objects.
        overlapped (tiles only): the all-gather goes chunk by chunk on a second stream behind the accumulation."""
        mine = sd.chromosome_shard(p, rank, world) if by_chromosome else p
        plan = secedo_amd.SimilarityMatrixPlan(local_rank)
        resident = plan.upload(mine, None, n_cells)  # the raw flat pileup, in HBM before the clock starts
        t0 = time.perf_counter()
        plan.prepare_resident(resident, n_cells, mfl, threads)
        torch.cuda.synchronize()
        prepare_s = time.perf_counter() - t0
        block_cells = 0  # the library's choice
        if by_chromosome:
            # shards are summed into one accumulator: same tile edge, same fixed-point scale on every rank
            block_cells, _ = sd.agree_on_shard_geometry(
                plan, lambda b: plan.prepare_resident(resident, n_cells, mfl, threads, b), world, red_dev)
        acc = plan.new_acc(pad_tiles_to=1 if by_chromosome else world)
        comm = torch.cuda.Stream() if overlapped else None
        my_tiles = (0, plan.num_tiles) if by_chromosome else sd.tile_range(plan.num_tiles, rank, world)

        def step():
            if not args.packed_resident:
                plan.prepare_resident(resident, n_cells, mfl, threads, block_cells)
            if not dist_on:
                plan.assign_finalize(acc, *rates, norm, out)  # one rank: every tile, then the normalisation
                return
            if by_chromosome:
                sd.chromosome_sharded_accumulate(plan, acc, *rates, world)
            elif overlapped:
                sd.sharded_accumulate_overlapped(plan, acc, *rates, rank, world, chunks=args.chunks, comm_stream=comm)
            else:
                sd.sharded_accumulate(plan, acc, *rates, rank, world)
            plan.finalize(acc, norm, out)

        for _ in range(args.warmup):
            step()
        sync()
        # exact integer work counters of one pass over this rank's tiles (one untimed launch of the whole range: the
        # chunked exchange leaves the counters of its last chunk only)
        plan.accumulate(acc, *rates, *my_tiles, overwrite=True)
        local_updates, local_pairs = plan.last_counts()
        sync()
        # EXACTLY K steps between barrier + synchronize on both sides, `--repeats` times over; the reported block
        # is the median one (max over ranks of each block first), the others give the spread
        blocks = []
        for _ in range(max(1, args.repeats)):
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            blocks.append(time.perf_counter() - t0)
        last_ms = plan.last_accumulate_ms()  # HIP events recorded by the library on the launch stream
        bt = torch.tensor(blocks, dtype=torch.float64, device=red_dev)
        if dist_on:
            dist.all_reduce(bt, op=dist.ReduceOp.MAX)
        blocks = sorted(float(v) for v in bt)
        elapsed = blocks[len(blocks) // 2]

        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        own = by_chromosome or rank == 0
        cnt = torch.tensor([local_updates, local_pairs, plan.num_entries if own else 0,
                            plan.num_reads if own else 0], dtype=torch.int64, device=red_dev)
        if dist_on:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        updates, pairs, kept_entries, reads = (int(v.item()) for v in cnt)
        return dict(by_chromosome=by_chromosome, overlapped=overlapped, plan=plan, resident=resident, acc=acc, my_tiles=my_tiles,
                    block_cells=block_cells, elapsed=float(t.item()), blocks=blocks, updates=updates, pairs=pairs,
                    kept_entries=kept_entries, reads=reads, local_updates=local_updates, last_ms=last_ms,
                    prepare_s=prepare_s)

    def run_config5():
        """BASELINE.json configs[4] as written: the matrix kept sharded by rows (NO all-gather: every rank
        accumulates the tiles that touch its rows, one scalar max-reduce, each normalises its row block) and fed
        to the distributed spectral step (row-block products, one all-reduce of N x 32 fp64 per product)."""
        plan = secedo_amd.SimilarityMatrixPlan(local_rank)
        resident = plan.upload(p, None, n_cells)
        t0 = time.perf_counter()
        plan.prepare_resident(resident, n_cells, mfl, threads)
        torch.cuda.synchronize()
        prepare_s = time.perf_counter() - t0
        acc = plan.new_acc()
        lo, hi = sd.row_range(n_cells, rank, world, plan.block_cells)
        ids = plan.tiles_of_rows(lo, hi)
        # exact work counters of the whole matrix: every tile counted once, by the rank that owns its row block
        nb = -(-n_cells // plan.block_cells)
        t_row = np.concatenate([np.full(nb - i, i, dtype=np.int64) for i in range(nb)])
        own = ids[(t_row[ids] * plan.block_cells >= lo) & (t_row[ids] * plan.block_cells < hi)]
        plan.accumulate_list(acc, *rates, own, overwrite=True)
        torch.cuda.synchronize()
        own_updates, own_pairs = plan.last_counts() if len(own) else (0, 0)
        state = {}

        def matrix_step():
            if not args.packed_resident:
                plan.prepare_resident(resident, n_cells, mfl, threads)
            state["rows"], state["lo"] = sd.sharded_rows(plan, acc, *rates, rank, world, norm)

        def spectral_step():
            state["eig"] = sd.sharded_eigenpairs(state["rows"], state["lo"], n_cells)

        def step():
            matrix_step()
            spectral_step()

        for _ in range(args.warmup):
            step()
        blocks = []
        for _ in range(max(1, args.repeats)):
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            blocks.append(time.perf_counter() - t0)
        phases = {}
        for name, fn in (("matrix_sharded_rows_ms", matrix_step), ("spectral_sharded_ms", spectral_step)):
            ts = []
            for _ in range(3):
                sync()
                t0 = time.perf_counter()
                fn()
                sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            phases[name] = sorted(ts)[1]
        bt = torch.tensor(blocks + [phases["matrix_sharded_rows_ms"], phases["spectral_sharded_ms"]],
                          dtype=torch.float64, device=red_dev)
        cnt = torch.tensor([own_updates, own_pairs, len(ids)], dtype=torch.int64, device=red_dev)
        if world > 1:
            dist.all_reduce(bt, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        blocks = sorted(float(v) for v in bt[:-2])
        elapsed = blocks[len(blocks) // 2]
        if rank == 0:
            vals, _, info = state["eig"]
            step_s = elapsed / args.steps
            updates = int(cnt[0].item())
            emit({
                "metric": "cell-pair x locus updates/sec (similarity matrix)", "value": updates / step_s,
                "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "int64", "data": "synthetic (SYNTH-v1, seed 42)",
                "config": {"workload": "%s: %d cells x %d loci, matrix kept sharded by rows (no all-gather) + distributed "
                                       "spectral step (BASELINE configs[4])" % (args.workload, n_cells, n_loci),
                           "updates_per_step": updates, "read_pairs_per_step": int(cnt[1].item()),
                           "tiles": plan.num_tiles, "tiles_accumulated_over_all_ranks": int(cnt[2].item()),
                           "block_cells": plan.block_cells, "normalization": norm,
                           "parallelism": "row blocks/%d: tiles touching own rows + scalar max all-reduce; spectral: "
                                          "row-block products + all-reduce of N x 32 fp64" % world,
                           "backend": backend_name, "world_size": world, "rccl_world_size": pg_world},
                "block_ms": [b / args.steps * 1e3 for b in blocks],
                "spread": (blocks[-1] - blocks[0]) / elapsed if elapsed else None,
                "phase_ms": {"matrix_sharded_rows": float(bt[-2].item()), "spectral_sharded": float(bt[-1].item())},
                "spectral": {"smallest_eigenvalues": [float(v) for v in vals[:4]], **info},
                "first_prepare_s": prepare_s, "step_includes_packing": not args.packed_resident,
                "roofline": None, "cpu_baseline": None,
            })

    backend_name = "none" if not dist_on else ("RCCL (nccl)" if args.backend == "nccl" else args.backend)
    pg_world = dist.get_world_size() if dist_on else 1   # read back from the process group, not from argv
    if world > 1 and args.workload == "C5" and not args.gathered:
        run_config5()
        dist.barrier()
        dist.destroy_process_group()
        return

    modes = []
    overlap_error = None
    if not dist_on:
        modes.append(run_mode(False))
    else:
        if args.only in ("both", "tiles"):
            modes.append(run_mode(False))
            if args.chunks > 1:
                # (the side-stream exchange runs under RCCL at world size 1 in tests/test_gpu_distributed.py; should it
                # raise here the line is still made from the other partitionings -- the ranks agree on that first: an
                # exception on ONE rank must not leave the others inside a collective of the next partitioning)
                mode_o = None
                try:
                    mode_o = run_mode(False, overlapped=True)
                except Exception as exc:  # noqa: BLE001
                    overlap_error = "%s: %s" % (type(exc).__name__, str(exc)[:300])
                ok = torch.tensor([0 if mode_o is None else 1], dtype=torch.int64, device=red_dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    modes.append(mode_o)
                elif overlap_error is None:
                    overlap_error = "another rank failed in the overlapped exchange"
        if args.only in ("both", "chromosomes") and p.n_chr >= 2:
            modes.append(run_mode(True))
    best = min(modes, key=lambda m: m["elapsed"])
    plan, acc, resident = best["plan"], best["acc"], best["resident"]
    by_chromosome, block_cells, my_tiles = best["by_chromosome"], best["block_cells"], best["my_tiles"]
    elapsed, updates = best["elapsed"], best["updates"]

    # parity of the LAST TIMED OUTPUT (outside the timed region): `out` still holds what the last timed step of
    # the last partitioning wrote; held against the compiled reference's sampled entries where a digest exists
    parity = None
    digest = reference_digest(args.workload, args.clustered, int(p.n_entries), rates, mfl, threads) if rank == 0 else None
    if digest is not None:
        ii = torch.from_numpy(digest["sample_i"].astype(np.int64)).to(out.device)
        jj = torch.from_numpy(digest["sample_j"].astype(np.int64)).to(out.device)
        err = np.abs(out[ii, jj].cpu().numpy() - digest["sample_v"])
        ref_max = float(digest["max_abs"])
        parity = {"checked": True, "against": "compiled reference (oracle/_ref), tests/golden/%s%s_reference_digest.npz"
                                              % (args.workload.lower(), "_clustered" if args.clustered else ""),
                  "samples": int(len(err)), "normwise_err": float(err.max() / ref_max), "tolerance": 1e-9,
                  "max_abs_matches": bool(abs(float(out.abs().max()) - ref_max) <= 1e-9 * ref_max),
                  "symmetric": bool(torch.equal(out, out.T)), "zero_diagonal": not bool(torch.any(torch.diagonal(out)))}
        parity["ok"] = bool(parity["normwise_err"] <= 1e-9 and parity["max_abs_matches"] and parity["symmetric"]
                            and parity["zero_diagonal"])

    # phase times of one step of the reported partitioning, events on the launch stream (torch's current one)
    phase = {}

    def repack():
        plan.prepare_resident(resident, n_cells, mfl, threads, block_cells)

    def refinalize():
        plan.finalize(acc, norm, out)

    for name, fn in (("pack_ms", repack), ("finalize_ms", refinalize)):
        ts = []
        if name == "finalize_ms":
            plan.accumulate(acc, *rates, *my_tiles, overwrite=True)
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
    evs, pair_samples = [], []
    for _ in range(min(args.steps, 10)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        plan.accumulate(acc, *rates, lo, hi, overwrite=True)
        b.record()
        evs.append((a, b))
        pm = plan.last_pair_kernel_ms()  # the pair kernel alone in this launch (HIP events of the library)
        if pm:
            pair_samples.append(pm)
    torch.cuda.synchronize()
    ev_ms = sorted(a.elapsed_time(b) for a, b in evs)
    kern_ms = ev_ms[len(ev_ms) // 2]
    pair_ms = sorted(pair_samples)[len(pair_samples) // 2] if pair_samples else None

    if rank == 0:
        # counter-derived inputs: HBM bytes per launch (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
        # this same command, calibrated as the microarch guide prescribes) and the SQ counters of the pair kernel
        key = args.workload if world == 1 else ""
        key += "_clustered" if args.clustered and key else ""
        traffic_rec = counters = None
        for name in ("traffic", "counters"):
            try:
                with open(os.path.join(ROOT, "profiles", "%s_%s.json" % (PROFILE_ROUND, name))) as fh:
                    rec = json.load(fh).get(key)
            except (OSError, ValueError):
                rec = None
            if name == "traffic":
                traffic_rec = rec
            else:
                counters = rec
        counters_live = False
        if world == 1 and not args.no_cpu_baseline and not args.no_live_counters:
            # (the full default command only: four child runs of about ten seconds each)
            live_t, live_c = live_counters(args, plan.pair_kernel, pair_ms if pair_ms else kern_ms)
            if live_t and live_c:
                traffic_rec, counters, counters_live = live_t, live_c, True
        E, L, N = plan.num_entries, plan.num_loci, n_cells
        local_updates = best["local_updates"]
        # this rank's launch: its tiles of the whole pileup, or all tiles of its chromosomes
        b_alg = (16 * local_updates + 6 * E + 4 * L + 16 * N * N if by_chromosome
                 else 16 * local_updates + 6 * E / world + 4 * L + 16 * N * N / world)
        b_alg_step = 16 * updates + 6 * best["kept_entries"] + 4 * n_loci + 16 * N * N  # the whole job
        step_s = elapsed / args.steps
        dom_ms = pair_ms if pair_ms else kern_ms
        dom_name = plan.pair_kernel if pair_ms else plan.pair_kernel + " (+ its reduction kernels: the accumulate phase)"
        peak, peak_src, peak_rec = lds_atomic_peak()
        # one LDS atomic per update in both pair kernels (ds_add_u32 into the count tile / ds_add_u64 into the int64 tile)
        achieved = local_updates / (dom_ms * 1e-3) / 1e9
        roofline = {
            # The binding limit (PMC: profiles/r03_pmc_*.txt): the updates are LDS atomics at random addresses of
            # the workgroup's tile; HBM carries 0.1 of the brief's per-update byte count and runs at < 0.25 of peak.
            "bound": "lds_atomic", "achieved": achieved, "peak": peak, "unit": "Gatomic/s",
            "frac": (achieved / peak) if peak else None,
            "peak_source": peak_src,
            "peak_bank_conflict_free": peak_rec.get("peak_bank_conflict_free_gatomic_per_s") if peak_rec else None,
            "kernel": dom_name, "kernel_ms": dom_ms, "updates_per_launch": local_updates,
            "accumulate_phase_ms": kern_ms, "kernel_ms_last_timed_step": best["last_ms"],
            # HBM bytes of the dominant kernel per launch, and the fraction of the HBM peak they amount to
            "traffic": traffic_rec["hbm_bytes_per_launch"] if traffic_rec else None,
            "traffic_source": (traffic_rec.get("source") if counters_live else
                               "profiles/%s_traffic.json (%s)" % (PROFILE_ROUND, traffic_rec.get("source", "rocprofv3 --pmc")))
            if traffic_rec else None,
            "hbm_frac_measured": (traffic_rec["hbm_bytes_per_launch"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)
            if traffic_rec else None,
            # SURVEY 8d's contract figure, NOT a fraction of anything physical: 16 B per update are charged to HBM
            # although the updates land in LDS (SURVEY 8d foresees values above 1 for such a design)
            "b_alg": {"algorithmic_bytes": b_alg, "gbps_over_accumulate_phase": b_alg / (kern_ms * 1e-3) / 1e9,
                      "b_alg_frac": b_alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                      "whole_step_b_alg_frac": b_alg_step / step_s / 1e9 / HBM_PEAK_GBPS / world},
        }
        if counters:
            # VALU issue: a wave64 VALU instruction occupies its SIMD's issue port for 2 cycles on gfx950 (guide)
            cyc = counters.get("GRBM_GUI_ACTIVE") or counters.get("kernel_cycles")
            if cyc and counters.get("SQ_INSTS_VALU"):
                roofline["valu_issue_frac"] = counters["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
            if counters.get("SQ_WAIT_ANY") and counters.get("SQ_WAVE_CYCLES"):
                roofline["wait_frac_of_wave_cycles"] = counters["SQ_WAIT_ANY"] / counters["SQ_WAVE_CYCLES"]
            if counters.get("SQ_LDS_BANK_CONFLICT") and counters.get("SQ_LDS_IDX_ACTIVE"):
                roofline["lds_bank_conflict_frac"] = counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"]
            if cyc and counters.get("SQ_LDS_IDX_ACTIVE"):
                # share of the kernel's CU-cycles (256 CUs) in which a CU's LDS pipeline has an instruction in hand
                roofline["lds_pipe_busy_frac"] = counters["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
            roofline["counters_source"] = (counters.get("source") if counters_live
                                           else "profiles/%s_counters.json" % PROFILE_ROUND)
        line = {
            "metric": "cell-pair x locus updates/sec (similarity matrix)",
            "value": updates * args.steps / elapsed,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic (SYNTH-v1, seed 42)",
            "config": {"workload": "%s: %d cells x %d loci%s" % (
                args.workload, n_cells, n_loci, " clustered" if args.clustered else ""),
                "entries": int(p.n_entries), "kept_entries": best["kept_entries"], "reads": best["reads"],
                "updates_per_step": updates, "read_pairs_per_step": best["pairs"],
                "block_cells": plan.block_cells, "tiles": plan.num_tiles,
                "normalization": norm, "max_fragment_length": mfl, "num_threads": threads,
                "parallelism": ("single GPU" if not dist_on else
                                "chromosomes/%d (packing + accumulation) + all-reduce of the int64 accumulator" % world
                                if by_chromosome else "tiles/%d (replicated packing) + all-gather of the int64 "
                                                      "accumulator%s" % (world, " in %d chunks on a second stream behind "
                                                                                "the accumulation" % args.chunks
                                                                         if best["overlapped"] else "")),
                "backend": backend_name, "world_size": world, "rccl_world_size": pg_world,
                # tiles mode: every rank packs the whole pileup, only the pair accumulation is divided
                "replicated_ms_per_step": phase.get("pack_ms") if world > 1 and not by_chromosome else None},
            # the K-step block, timed `repeats` times: value / ms_per_step are the median block's
            "repeats": len(best["blocks"]),
            "block_ms_per_step": [b / args.steps * 1e3 for b in best["blocks"]],
            "spread": (best["blocks"][-1] - best["blocks"][0]) / elapsed if elapsed else None,
            "parity_checked": bool(parity and parity["ok"]),
            "parity": parity,
            "partitionings": {("chromosomes+all-reduce" if m["by_chromosome"] else
                               "tiles+all-gather in %d chunks behind the accumulation" % args.chunks if m["overlapped"]
                               else "tiles+all-gather"):
                              {"ms_per_step": m["elapsed"] / args.steps * 1e3,
                               "updates_per_s": m["updates"] * args.steps / m["elapsed"]} for m in modes},
            "overlapped_exchange_error": overlap_error,
            "wall_s_full_matrix": step_s,
            "dense_equivalent_cell_pair_locus_slots_per_s": n_cells * (n_cells - 1) / 2 * n_loci / step_s,
            "step_includes_packing": not args.packed_resident,
            "packing": "device" if plan.used_device_packing else "host",
            "phase_ms": {"pack": phase["pack_ms"], "accumulate": kern_ms, "finalize": phase["finalize_ms"]},
            "first_prepare_s": best["prepare_s"],
            "scale_log2": plan.scale_log2,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            # the drop-in call itself: host buffers in, host matrix out (H2D, pack, accumulate, normalise,
            # D2H) -- PCIe-inclusive, never `value`
            # (the library keeps the call's device buffers for the next call: first and repeated call)
            t0 = time.perf_counter()
            secedo_amd.compute_similarity_matrix(p, n_cells, mfl, None, *rates, threads, "", norm)
            line["one_shot_first_call_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            secedo_amd.compute_similarity_matrix(p, n_cells, mfl, None, *rates, threads, "", norm)
            line["one_shot_host_call_s"] = time.perf_counter() - t0
            # ... and through the C++ shim with the reference's signature (include/secedo_simmat.hpp) from a
            # vector<vector<PosData>>, as the reference's caller holds the pileup: flatten + H2D + step +
            # D2H straight into the returned matrix. A child process (tests/cpp/shim_test.cpp, built by make).
            shim = os.path.join(ROOT, "secedo_amd", "csrc", "build", "shim_test")
            if os.path.exists(shim) and n_cells <= 16383:
                try:
                    r = subprocess.run([shim, "--synth", str(n_cells), str(n_loci), str(n_chr),
                                        str(300 if args.clustered else gap), repr(prob), "2"],
                                       capture_output=True, text=True, timeout=600)
                    line["cpp_dropin_call"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 \
                        else {"error": r.stderr[-300:]}
                except (subprocess.SubprocessError, ValueError, IndexError) as e:
                    line["cpp_dropin_call"] = {"error": str(e)[:300]}
            line["cpu_baseline"] = cpu_baseline(p, n_cells, mfl, rates, threads, 5e8)
            if digest is not None and "reference_seconds" in digest.files:
                # the WHOLE workload through the compiled reference, once, in the build container (8 vCPUs), when
                # the digest above was made (oracle/gen_golden.py): not a measurement of this run
                line["cpu_baseline"]["reference_full_workload_s_in_build_container"] = float(digest["reference_seconds"])
        emit(line)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
