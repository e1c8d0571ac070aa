#!/bin/bash
# kernel_regs.sh -- VGPR / spill counts of the accumulate_tiles variants in the current build
cd "$(dirname "$0")/../secedo_amd/csrc/build" || exit 1
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading simmat_kernels.o >/dev/null 2>&1
f=$(ls | grep "simmat_kernels.o.0.hipv4")
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | grep -E "\.name:|vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed" | paste - - - - - | grep accumulate | sed 's/_ZN6secedo12_GLOBAL__N_1//' | awk '{print $2, "scratch", $4, "sgpr_spill", $6, "vgpr", $8, "vgpr_spill", $10}'
rm -f simmat_kernels.o.0.*
