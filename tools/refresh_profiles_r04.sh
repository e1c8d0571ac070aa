#!/bin/bash
# refresh_profiles_r04.sh -- the artefacts of round 4 under profiles/ (run on the GPU box; output in
# gpurun_out/r04, copied to profiles/r04_* by hand). One gpurun call, about six minutes.
#   1. the driver's own command, `python bench.py` (C3 = 8000 cells x 100K loci), and the LDS-atomic peak
#   2. rocprofv3 --kernel-trace --stats over the same command (program directly after --)
#   3. FETCH_SIZE / WRITE_SIZE passes over it (separate --pmc runs) + the calibration of both counters
#   4. SQ / LDS counters of the accumulate kernels -> pmc_C3.txt and counters.json (what bench.py's roofline reads)
#   5. the clustered-loci variants (gap_max = 300): bench lines + kernel stats of C3 and C2, SQ counters of C3
#   6. C2 and C5 bench lines
set -e
export TMPDIR=/tmp
export SECEDO_BENCH_NO_CHILD=1   # (set again to empty for the plain bench runs below: they measure the LDS peak live)
R=$GRAFT_REPO_ROOT/gpurun_out/r04
rm -rf $R; mkdir -p $R
tools/lds_atomic_bench.bin --json > $R/lds_atomic_peak.json
# which unit the microbenchmark saturates: one counter pass over it, the binary directly behind --
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/pmc_ldsbench -- tools/lds_atomic_bench.bin --json > /dev/null 2>&1
python tools/pmc_table.py $R/pmc_ldsbench > $R/lds_atomic_bench_pmc.txt
rm -rf $R/pmc_ldsbench
SECEDO_BENCH_NO_CHILD= python bench.py > $R/C3_bench.json 2> $R/C3_bench.err
echo "C3 bench done"; head -c 400 $R/C3_bench.json; echo
stats() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$name -- python bench.py --no-cpu-baseline --repeats 2 "$@" > $R/prof_$name.log 2>&1
  cp $(find $R/prof_$name -name "*kernel_stats.csv" | head -1) $R/${name}_kernel_stats.csv
  python tools/kstats.py $R/prof_$name > $R/${name}_kernel_stats_readable.txt
  rm -rf $R/prof_$name
  echo "$name kernel stats done"
}
stats C3
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/calib_$c -- tools/fetch_calib.bin > $R/calib_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $R/pmc_$c -- python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > $R/pmc_$c.log 2>&1
done
python tools/traffic_json.py $R > $R/traffic.json
echo "traffic done"
P="--steps 3 --warmup 1 --repeats 1"
tools/pmc.sh r04_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" $P > $R/pmc_C3_a.txt
tools/pmc.sh r04_b "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" $P > $R/pmc_C3_b.txt
tools/pmc.sh r04_c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" $P > $R/pmc_C3_c.txt
cat $R/pmc_C3_a.txt $R/pmc_C3_b.txt $R/pmc_C3_c.txt > $R/pmc_C3.txt
echo "pmc done"
# clustered loci
SECEDO_BENCH_NO_CHILD= python bench.py --clustered --no-cpu-baseline > $R/C3_clustered_bench.json 2> $R/C3_clustered_bench.err
SECEDO_BENCH_NO_CHILD= python bench.py --workload C2 --clustered --no-cpu-baseline > $R/C2_clustered_bench.json 2> $R/C2_clustered_bench.err
stats C3_clustered --clustered --steps 5
stats C2_clustered --workload C2 --clustered
tools/pmc.sh r04_ca "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" --clustered $P > $R/pmc_C3_clustered_a.txt
tools/pmc.sh r04_cb "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" --clustered $P > $R/pmc_C3_clustered_b.txt
cat $R/pmc_C3_clustered_a.txt $R/pmc_C3_clustered_b.txt > $R/pmc_C3_clustered.txt
python tools/counters_json.py $R > $R/counters.json
cat $R/counters.json
SECEDO_BENCH_NO_CHILD= python bench.py --workload C2 --no-cpu-baseline > $R/C2_bench.json 2> $R/C2_bench.err
SECEDO_BENCH_NO_CHILD= python bench.py --workload C5 --no-cpu-baseline --steps 5 > $R/C5_bench.json 2> $R/C5_bench.err
# the N > 1 step under a world-size-1 RCCL group (every collective issued), and the one-shot C++ call on 1 / 2 / 4 lanes
SECEDO_BENCH_NO_CHILD= python bench.py --group --no-cpu-baseline > $R/C3_group_rccl_world1_bench.json 2> $R/C3_group.err
for g in "" "--gpus 0,0" "--gpus 0,0,0,0"; do
  tag=$(echo "lanes$g" | tr -d ' ,-' | sed 's/gpus//')
  SECEDO_ONE_SHOT_TRACE=1 secedo_amd/csrc/build/shim_test $g --synth 8000 100000 22 30000 0.03 2 > $R/shim_C3_$tag.json 2> $R/shim_C3_$tag.trace
done
python tools/divide_cluster_demo.py C3 > $R/divide_cluster_demo_C3.json 2> /dev/null
python tools/divide_cluster_demo.py C2 > $R/divide_cluster_demo_C2.json 2> /dev/null
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_r04_* $R/pmc_FETCH_SIZE $R/pmc_WRITE_SIZE $R/calib_FETCH_SIZE $R/calib_WRITE_SIZE 2>/dev/null || true
echo "all done"
