#!/bin/bash
# poison_run.sh -- the GPU parity tests and a fuzz run with SECEDO_POISON=2: every device allocation of the
# library starts as 0xA5 bytes and the packing's scratch and outputs are refilled before every prepare, so a
# kernel that reads a word nobody wrote fails every time instead of once in a while (how the stale word behind
# blk_off in k_fix_locus_rel showed itself only on fresh handles). One gpurun call; output in gpurun_out/poison.
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/poison
mkdir -p $out
export SECEDO_POISON=2
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_filter.py -x -q -s > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -3 $out/tests.log
timeout -k 10 400 python tools/fuzz_parity.py 120 11 > $out/fuzz.log 2>&1
echo "fuzz exit $?" >> $out/fuzz.log
tail -2 $out/fuzz.log
