#!/bin/bash
# refresh_profiles.sh -- regenerates the artefacts kept under profiles/ (run on the GPU box; results in gpurun_out/refresh)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/refresh
rm -rf $R; mkdir -p $R
python bench.py --steps 20 --warmup 3 > $R/C2_bench.json 2> $R/C2_bench.err
echo "C2 bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_C2 -- python bench.py --steps 20 --warmup 3 > $R/prof_C2.log 2>&1
cp $(find $R/prof_C2 -name "*kernel_stats.csv" | head -1) $R/C2_kernel_stats.csv
python tools/kstats.py $R/prof_C2 > $R/C2_kernel_stats_readable.txt
echo "C2 kernel stats done"
python bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline > $R/C3_bench.json 2> $R/C3_bench.err
python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline > $R/C5_bench.json 2> $R/C5_bench.err
python bench.py --workload C2 --clustered --steps 10 --warmup 2 --no-cpu-baseline > $R/C2_clustered_bench.json 2> $R/C2c.err
python bench.py --workload C2 --packed-resident --steps 20 --warmup 3 --no-cpu-baseline > $R/C2_resident_bench.json 2> $R/C2r.err
python bench.py --workload C3 --packed-resident --steps 5 --warmup 2 --no-cpu-baseline > $R/C3_resident_bench.json 2> $R/C3r.err
echo "benches done"
for wl in C2 C3; do
  steps=10; [[ $wl == C3 ]] && steps=3
  tools/pmc.sh ${wl}_fetch "FETCH_SIZE" --workload $wl --steps $steps --warmup 1 > $R/pmc_${wl}_fetch.txt
  tools/pmc.sh ${wl}_write "WRITE_SIZE" --workload $wl --steps $steps --warmup 1 > $R/pmc_${wl}_write.txt
done
echo "traffic done"
tools/pmc.sh C3_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" --workload C3 --steps 3 --warmup 1 > $R/pmc_C3_a.txt
tools/pmc.sh C3_b "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" --workload C3 --steps 3 --warmup 1 > $R/pmc_C3_b.txt
tools/pmc.sh C3_c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" --workload C3 --steps 3 --warmup 1 > $R/pmc_C3_c.txt
echo "pmc done"
