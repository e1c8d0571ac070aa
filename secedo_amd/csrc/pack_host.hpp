// pack_host.hpp -- host-side read assembly, flush schedule and tile packing.
//
// Turns the flat pileup of include/secedo_simmat.h into the device layout of the accumulate
// kernel (DESIGN.md "Data layout in HBM"). This is where the reference's sequential semantics
// live (all cheap, O(entries)):
//   * read assembly and the paired-end duplicate rule  (reference: similarity_matrix.cpp:376-403)
//   * the flush schedule with threshold 4*num_threads   (:348-373)
//   * re-opening of a read id after its read was flushed (:368-371 + :379-382)
//   * the tail rule: reads still live at the end of a chromosome never act as the earlier read
//     of a pair (:407-408 clears the live list before the loop at :412-418 runs)
// The quadratic work (pair enumeration, accumulation) and the normalisation run on the GPU.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace secedo {

struct FlatPileupView {
    const uint32_t *chr_locus_off = nullptr;
    uint32_t n_chr = 0;
    const uint32_t *locus_pos = nullptr;
    const uint64_t *locus_entry_off = nullptr;
    const uint32_t *read_ids = nullptr;
    const uint16_t *id_base16 = nullptr;  // exactly one of id_base16 / id_base32
    const uint32_t *id_base32 = nullptr;
    const uint32_t *group_id_to_pos = nullptr;
    uint32_t n_groups = 0;

    uint32_t n_loci() const { return chr_locus_off[n_chr]; }
    uint64_t n_entries() const { return locus_entry_off[n_loci()]; }
    uint32_t id_base(uint64_t e) const { return id_base16 ? id_base16[e] : id_base32[e]; }
};

// One kept pileup entry in full: 16 bytes, everything a pair needs unless a 16-locus window
// overflows. "Window" = the loci before / after the entry's locus (locus INDEX distance within
// the pileup, not base pairs).
struct Entry {
    uint32_t meta;    // bits 0-15 cell (matrix row), 16-17 base, 18 tail, 19 prev_ovf, 20 next_ovf
    uint32_t masks;   // bit t (t < 16): the read also has a kept entry at locus - 1 - t;
                      // bit 16 + t: ... at locus + 1 + t
    uint32_t bases;   // bit t: low bit of the base at locus + 1 + t; bit 16 + t: its high bit
    uint32_t locus;   // global locus index
};

constexpr uint32_t kMetaBaseShift = 16;
constexpr uint32_t kMetaTail = 1u << 18;
constexpr uint32_t kMetaPrevOvf = 1u << 19;
constexpr uint32_t kMetaNextOvf = 1u << 20;
constexpr uint32_t kWindow = 16;

// The compact form the accumulate kernel streams (4 bytes per entry; its low 16 bits are what is
// staged in LDS for the column side of a tile):
//   bits 0-6  cell - block * B      bits 7-8 base      bit 9 tail
//   bit 10    multi: the read has more than one kept entry (else every pair shares this locus only)
//   bit 11    wide: the read reaches beyond the 8-locus windows of mask32 -> use the full Entry
//   bits 16-31 locus - first locus of the entry's locus range
constexpr uint32_t kC_BaseShift = 7;
// col32: an entry as the column side of accumulate_counts stages it in LDS -- byte offset of the cell's column in a
// tile row (cell in block * 4) in the low half, the base in the byte above: address and base test are one
// sub-dword operand each. (Made of entry32 when a range is staged; until round 3 the packing wrote a second copy
// of every entry in this form.)
constexpr uint32_t col32_of(uint32_t cell_in_block, uint32_t base) { return (cell_in_block << 2) | (base << 16); }
constexpr uint32_t kC_Tail = 1u << 9;
constexpr uint32_t kC_Multi = 1u << 10;
constexpr uint32_t kC_Wide = 1u << 11;
constexpr uint32_t kNarrowWindow = 8;

// LDS staging limits of the accumulate kernel (simmat_kernels.hpp supplies them per tile size)
struct StageGeometry {
    uint32_t cap_entries_plain, cap_loci_plain;    // compact records only, int64 tile
    uint32_t cap_entries_masks, cap_loci_masks;    // compact records + mask32 staged as well
    uint32_t cap_entries_counts, cap_loci_counts;  // compact records only, 2 x 16-bit count tile
    double masks_threshold;                        // stage mask32 when this fraction of entries is multi
};

// 2 x 16-bit pair counters per cell pair are safe while no cell pair can collect this many pairs
constexpr uint64_t kCountTileLimit = 65536;

struct PackedPileup {
    uint32_t num_cells = 0;
    uint32_t block_cells = 0;   // B: cells per tile edge
    uint32_t num_blocks = 0;    // ceil(num_cells / B)
    uint32_t num_loci = 0;
    uint64_t num_entries = 0;   // kept entries
    uint64_t num_reads = 0;     // live reads (segments)
    uint64_t raw_entries = 0;   // entries of the input pileup
    uint64_t pair_bound = 0;    // max over cells of sum_l n_cell(l)^2 (Cauchy-Schwarz bound)
    std::vector<uint64_t> cell_sq;  // the per-cell sums themselves
    uint64_t multi_entries = 0; // entries of reads with more than one kept entry
    uint32_t n_wide = 0;        // kept entries whose read reaches beyond their 8-locus windows (kC_Wide in entry32)
    uint32_t max_read_entries = 0; // kept entries of the longest read: no read pair shares more loci
    bool any_window_overflow = false;
    bool stage_masks = false;
    bool count_tile = false;    // the accumulate kernel may use the 2 x 16-bit count tile

    // entries sorted by (cell block, locus, input order); blk_off[b * (L+1) + l] is the first
    // entry of block b at locus l, blk_off[b * (L+1) + L] the end of block b
    std::vector<uint32_t> blk_off;
    std::vector<uint32_t> entry32;     // compact form
    std::vector<uint32_t> mask32;      // prev8 | next8 << 8 | next_b0 << 16 | next_b1 << 24
    std::vector<Entry> entry;          // full form (pairs of two multi reads beyond mask32)
    std::vector<uint32_t> entry_read;  // per entry: index of its live read (slow path only)
    // locus ranges the accumulate kernel stages through LDS: range r covers loci
    // [range_off[r], range_off[r+1]); no block has more than cap_entries entries in a range and no
    // range more than cap_loci loci -- except single-locus ranges whose locus alone exceeds the cap
    // (the kernel pairs those straight from HBM)
    std::vector<uint32_t> range_off;
    uint32_t cap_entries = 0, cap_loci = 0;
    // per live read: its kept entries in locus order (slow path: windows overflowed)
    std::vector<uint32_t> read_off;   // num_reads + 1
    std::vector<uint32_t> read_locus; // num_entries
    std::vector<uint8_t> read_base;   // num_entries
};

// Returns an empty string on success, else an error message (invalid input).
// block_cells: 64 or 128, or 0 = choose: 128-cell tiles amortise the per-batch work best, but
// only 64-cell tiles leave LDS room for the window masks that clustered loci need.
// allow_count_tile = false forces the int64 tile (diagnostics)
std::string pack_pileup(const FlatPileupView &in, uint32_t num_cells, uint32_t max_fragment_length,
                        uint32_t num_threads, uint32_t block_cells,
                        StageGeometry (*geometry)(uint32_t block_cells), bool allow_count_tile,
                        PackedPileup *out);

}  // namespace secedo
