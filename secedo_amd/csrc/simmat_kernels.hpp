// simmat_kernels.hpp -- launch interface of simmat_kernels.hip (device pointers only).
#pragma once

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace secedo {

struct LlrModelDev {  // LlrModel of llr_table.hpp, by value into the kernel
    double ln_u1, ln_v1, ln_u2, ln_v2, ln_w1, ln_z1, ln_w2, ln_z2;
};

struct AccumulateArgs {
    // packed pileup (pack_host.hpp), all in HBM
    const uint32_t *blk_off;   // num_blocks * stride
    uint32_t stride;           // num_loci + 1
    uint32_t num_loci;
    const uint4 *entry_a;      // EntryA
    const uint4 *entry_b;      // EntryB
    const uint32_t *read_off;
    const uint32_t *read_locus;
    const uint8_t *read_base;
    // tiles: upper-triangular block pairs, row-major
    const uint16_t *tile_row;
    const uint16_t *tile_col;
    uint32_t tile_begin;       // first tile of this launch
    uint32_t n_chunks;         // locus chunks per tile (workgroups per tile)
    uint32_t chunk_loci;       // loci per chunk
    // log-likelihood ratios, fixed point
    const long long *lut;      // 65 x 65, row = x_s
    LlrModelDev model;
    int scale_log2;
    // outputs
    int64_t *acc;              // tile-major: [tile][B*B]
    unsigned long long *counters;  // [0] incidences examined, [1] read pairs accumulated
};

hipError_t launch_accumulate(const AccumulateArgs &args, uint32_t block_cells, uint32_t n_tiles,
                             hipStream_t stream);

// mode 0..2 = SECEDO_NORM_*, 3 = raw D
hipError_t launch_finalize(const int64_t *acc, uint32_t n, uint32_t nb, uint32_t block_cells,
                           int scale_log2, int mode, unsigned long long *d_max_bits, double *out,
                           hipStream_t stream);

}  // namespace secedo
