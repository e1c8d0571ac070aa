#!/bin/bash
# refresh_profiles_r02.sh -- the artefacts of round 2 under profiles/ (run on the GPU box; output in
# gpurun_out/r02, copied to profiles/ by hand). One gpurun call.
#   1. the driver's own command, `python bench.py` (C3 = 8000 cells x 100K loci)
#   2. rocprofv3 --kernel-trace --stats over the same command (program directly after --)
#   3. FETCH_SIZE / WRITE_SIZE passes over it (separate --pmc runs) + the calibration of both counters
#   4. SQ / LDS counters of the accumulate kernels (accumulate_counts, correct_tiles)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r02
rm -rf $R; mkdir -p $R
python bench.py > $R/C3_bench.json 2> $R/C3_bench.err
echo "C3 bench done"; cat $R/C3_bench.json | head -c 600; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_C3 -- python bench.py --no-cpu-baseline > $R/prof_C3.log 2>&1
cp $(find $R/prof_C3 -name "*kernel_stats.csv" | head -1) $R/C3_kernel_stats.csv
python tools/kstats.py $R/prof_C3 > $R/C3_kernel_stats_readable.txt
rm -rf $R/prof_C3
echo "C3 kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/calib_$c -- tools/fetch_calib.bin > $R/calib_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $R/pmc_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/pmc_$c.log 2>&1
done
python tools/traffic_json.py $R > $R/traffic.json
cat $R/traffic.json
echo "traffic done"
tools/pmc.sh r02_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" --steps 3 --warmup 1 > $R/pmc_C3_a.txt
tools/pmc.sh r02_b "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" --steps 3 --warmup 1 > $R/pmc_C3_b.txt
tools/pmc.sh r02_c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" --steps 3 --warmup 1 > $R/pmc_C3_c.txt
cat $R/pmc_C3_a.txt $R/pmc_C3_b.txt $R/pmc_C3_c.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_r02_* $R/pmc_FETCH_SIZE $R/pmc_WRITE_SIZE $R/calib_FETCH_SIZE $R/calib_WRITE_SIZE 2>/dev/null || true
echo "pmc done"
