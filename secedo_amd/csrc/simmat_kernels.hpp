// simmat_kernels.hpp -- launch interface of simmat_kernels.hip (device pointers only).
#pragma once

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include "pack_host.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace secedo {

struct LlrModelDev {  // LlrModel of llr_table.hpp, by value into the kernel
    double ln_u1, ln_v1, ln_u2, ln_v2, ln_w1, ln_z1, ln_w2, ln_z2;
};

// LDS staging geometry of accumulate_tiles per tile size: column-side entries and loci per locus
// range, without / with the 8-locus window masks staged too. pack_host.cpp cuts the locus ranges
// to these limits. LDS per workgroup: B=64: 32 KiB tile + 12/28 KiB; B=128: 128 KiB tile + 24/28 KiB.
constexpr uint32_t kCapJ64 = 4096, kCapL64 = 2046, kCapJ64M = 2048, kCapL64M = 2046, kCapJ64C = 4096, kCapL64C = 2046;
constexpr uint32_t kCapJ128 = 4096, kCapL128 = 2046, kCapJ128M = 4096, kCapL128M = 2046;
// the 64 KiB count tile leaves room for longer ranges. Entries per range, C3 pair kernel with one range per segment
// (pack_device.hip): 6144 2.60 ms, 8192 2.34, 10240 2.33, 12288 2.24 (a range's two barriers, its staging and
// its prefetch are paid per range; 14336 needs the rings halved and 116 VGPRs: 2.5)
constexpr uint32_t kCapJ128C = 12288, kCapL128C = 8190;
constexpr double kMasksThreshold = 0.05;  // stage the masks when > 5 % of the entries are multi-locus

// Everything only the rare paths touch lives in HBM behind one pointer, so that the kernel's
// by-value argument block stays in SGPRs (passing it by reference to a non-inlined device
// function would put it in scratch memory and every hot-loop access with it).
struct SlowPathArgs {
    const uint4 *entry;        // full entries (Entry)
    const uint32_t *entry_read;
    const uint32_t *read_off;
    const uint32_t *read_locus;
    const uint8_t *read_base;
    const long long *lut;      // 129 x 129 (LUT_DIM), row = x_s
    LlrModelDev model;
    int scale_log2;
    // Read pairs that share more than 128 loci (beyond the table): with a list here the kernels add NOTHING for the
    // joint term of such a pair and note {cell, cell, x_s, x_d} instead; the host evaluates the term as the reference
    // does and adds it afterwards (simmat_api.cpp). nullptr: the closed form on the device (SECEDO_LLR_EXACT).
    uint4 *beyond_list = nullptr;
    uint32_t *beyond_count = nullptr;   // entries noted (may exceed the capacity: then the list is incomplete)
    uint32_t beyond_cap = 0;
};

struct AccumulateArgs {
    // packed pileup (pack_host.hpp), all in HBM
    const uint32_t *blk_off;   // num_blocks * stride
    uint32_t stride;           // num_loci + 1
    const uint32_t *entry32;   // compact entries
    const uint32_t *mask32;    // 8-locus windows
    const uint4 *entry;        // full entries (Entry), for locus ranges too deep to stage
    const uint32_t *range_off; // num_ranges + 1 locus boundaries
    uint32_t num_ranges;
    const SlowPathArgs *slow;  // in HBM
    // tiles: upper-triangular block pairs, row-major
    const uint16_t *tile_row;
    const uint16_t *tile_col;
    uint32_t tile_begin;       // first tile of this launch (tile_ids == nullptr)
    const uint32_t *tile_ids;  // or: the launch's tiles by global index (a rank's row block)
    uint32_t n_tiles;          // tiles of this launch
    uint32_t n_workgroups;     // = tile_wg_begin[n_tiles]
    const uint32_t *wg_tile;   // workgroup -> tile (relative to tile_begin)
    const uint32_t *tile_wg_begin;  // n_tiles + 1: first workgroup of each tile (its chunks follow)
    uint32_t debug;            // ablation switches for profiling (0 in production)
    // log-likelihood ratios, fixed point
    const long long *lut;      // 129 x 129 (LUT_DIM), row = x_s
    // outputs
    int64_t *acc;              // tile-major: [tile][B*B]
    void *slab;                // one B*B tile (u32 counts or int64) per workgroup of the launch
    unsigned long long *counters;  // [0] incidences examined, [1] read pairs accumulated
    // the sparse-loci path (accumulate_counts + correct_tiles): the flagged entries, compact (build_flagged_lists)
    const uint32_t *flag_grp = nullptr;     // num_blocks * stride: flagged entries before each (block, locus) group
    const uint4 *flag_rec = nullptr;        // their full entries ...
    const uint32_t *flag_idx = nullptr;     // ... and entry indices
    int group_hint = 4;                     // GROUP of accumulate_counts by the entries per (cell block, locus)
    bool overwrite = false;                 // acc[tiles of the launch] = result (no need to zero them first)
    bool masks_kernel = false;              // staged masks: accumulate_masks (+ wide_pairs) instead of accumulate_tiles
    uint32_t masks_slot_asm = 1;            // accumulate_masks: the hand-written pair slots (SECEDO_MASKS_SLOT_ASM=0: off)
    const uint32_t *mk_y = nullptr, *mk_xcol = nullptr, *mk_xrow = nullptr;  // ... the entries' words (masks_words)
    const uint32_t *wide_off = nullptr;     // num_blocks + 1: the C_WIDE entries per cell block ...
    const uint32_t *wide_list = nullptr;    // ... their entry indices (null: none)
    // counts path, one workgroup per tile (counts_split(n_tiles) == 1), all tiles in one launch: max(0, max D) of
    // the stored tiles by atomicMax, as launch_tile_max leaves it (zeroed by the caller)
    unsigned long long *max_bits = nullptr;
    double max_scale = 0.0;
};

// true when the count-tile variants run accumulate_counts + correct_tiles (the default; SECEDO_PAIR_MODE=0
// selects the flattening kernel accumulate_tiles instead, for A/B measurements)
bool counts_path_enabled();
// workgroups per tile of correct_tiles for a launch of n_tiles
uint32_t counts_split(uint32_t n_tiles);
// The entries whose read is flagged (never flushed, or covering further loci), compacted in entry order --
// which is (cell block, locus) order -- for correct_tiles: pre[n_entries + 1] (exclusive prefix of the flag,
// scratch), grp[n_off] = pre at the n_off group offsets blk_off[], rec[] / idx[] = the flagged entries' records
// and indices (room for n_entries each). scan_tmp: at least flagged_scan_bytes(n_entries). Depends on the packed
// pileup only.
size_t flagged_scan_bytes(uint32_t n_entries);
hipError_t build_flagged_lists(const uint32_t *entry32, const uint4 *entry, uint32_t n_entries,
                               const uint32_t *blk_off, size_t n_off, void *scan_tmp, size_t scan_tmp_bytes,
                               uint32_t *pre, uint32_t *grp, uint4 *rec, uint32_t *idx, hipStream_t stream);

StageGeometry stage_geometry(uint32_t block_cells);
// The entries flagged C_WIDE (their read reaches beyond the 8-locus windows), listed per cell block for
// accumulate_masks' operands: per packed entry the base planes and the flag word in column and in row form, made once
// per prepare from entry32 / mask32
hipError_t masks_words(const uint32_t *entry32, const uint32_t *mask32, uint32_t n_entries, uint32_t *y, uint32_t *xcol,
                       uint32_t *xrow, hipStream_t stream);
// accumulate_masks' second kernel: wide_count leaves cnt[num_blocks], off[num_blocks + 1] (exclusive scan; the total
// in off[num_blocks]) and the fill cursors cur[num_blocks]; wide_fill writes the entry indices (room for the total).
hipError_t wide_count(const uint32_t *entry32, const uint32_t *blk_off, uint32_t stride, uint32_t num_blocks,
                      uint32_t *cnt, uint32_t *off, uint32_t *cur, hipStream_t stream);
hipError_t wide_fill(const uint32_t *entry32, const uint32_t *blk_off, uint32_t stride, uint32_t num_blocks,
                     uint32_t *cur, uint32_t *list, hipStream_t stream);

// workspace of one accumulate launch: a tile per workgroup (plain-store flush, then reduce_slabs)
size_t accumulate_slab_bytes(uint32_t block_cells, bool count_tile, uint32_t n_workgroups);

// stage_masks / count_tile: the kernel variants, see accumulate_tiles. count_tile requires fewer
// than 65536 pairs per cell pair (PackedPileup::pair_bound) and !stage_masks; stage_masks exists
// for 64-cell tiles only. The accumulator must be zeroed by the caller: the flush is additive.
// side: the stream (with two events) the flagged entries' lists are built on beside the pair kernel, or null
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    // work for the side stream that the launch issues right AFTER the pair kernel (so that the host's enqueueing of
    // it does not delay that kernel) and whose `join` the second kernel waits for; null: nothing pending
    hipError_t (*deferred)(void *ctx) = nullptr;
    void *deferred_ctx = nullptr;
};
// mid: when non-null, recorded on `stream` between the pair kernel and what follows it (the duration of the
// dominant kernel by itself: secedo_simmat_last_pair_kernel_ms)
hipError_t launch_accumulate(const AccumulateArgs &args, uint32_t block_cells, bool stage_masks,
                             bool count_tile, uint32_t n_tiles, hipStream_t stream, const SideStream *side = nullptr,
                             hipEvent_t mid = nullptr);

// mode 0..2 = SECEDO_NORM_*, 3 = raw D
hipError_t launch_finalize(const int64_t *acc, const uint16_t *tile_row, const uint16_t *tile_col, uint32_t n_tiles,
                           uint32_t n, uint32_t block_cells, int scale_log2, int mode,
                           unsigned long long *d_max_bits, uint32_t row_begin, uint32_t row_end, double *out,
                           hipStream_t stream, bool keep_max = false);
// d_max_bits = bits of max(0, max D) over the listed tiles (tile_ids == nullptr: all n_tiles)
// acc[index[k]] += value[k] for k < n (the joint terms of read pairs that share more than 128 loci)
hipError_t launch_add_terms(int64_t *acc, const unsigned long long *index, const long long *value, uint32_t n,
                            hipStream_t stream);
hipError_t launch_tile_max(const int64_t *acc, const uint16_t *tile_row, const uint16_t *tile_col,
                           const uint32_t *tile_ids, uint32_t n_tiles, uint32_t block_cells, int scale_log2,
                           unsigned long long *d_max_bits, hipStream_t stream);

}  // namespace secedo
