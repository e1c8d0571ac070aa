"""`python bench.py --gpus N` must start its own N ranks (VERDICT r01 "next" 2): the parent process, which has
not touched the GPU, runs torch.distributed.run; the ranks rendezvous on 127.0.0.1 and share the work out.
Here without a GPU, through bench.py's --dry-run (gloo; everything the ranks do before the first HIP call)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)  # as the driver starts it: no launcher around it
    env.pop("RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                          timeout=600, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_that_cover_the_work_once():
    line = run_bench("--gpus", "2", "--dry-run", "--workload", "C2")
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["backend"] == "gloo"
    assert sum(line["shard_entries"]) == line["entries"] and min(line["shard_entries"]) > 0
    assert sum(line["shard_chromosomes"]) == line["chromosomes"] == 22
    assert sum(line["shard_tiles"]) == line["tiles"] == 36


def test_bench_single_rank_needs_no_launcher():
    line = run_bench("--dry-run", "--workload", "C1")
    assert line["n_gpus"] == 1 and line["shard_entries"] == [line["entries"]]


def test_bench_default_workload_is_the_headline_configuration():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args([])
    assert args.workload == "C3" and args.gpus == 1  # BASELINE.json configs[2]: 8000 cells x 100K loci
