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

// One kept pileup entry, 2 x 16 bytes. "Window" = the 32 loci before / after the entry's locus
// (locus INDEX distance within the pileup, not base pairs).
struct EntryA {          // read by every pair
    uint32_t meta;       // bits 0-15 cell (matrix row), 16-17 base, 18 tail, 19 prev_ovf, 20 next_ovf
    uint32_t prev_mask;  // bit t: the read also has a kept entry at locus - 1 - t
    uint32_t next_mask;  // bit t: the read also has a kept entry at locus + 1 + t
    uint32_t locus;      // global locus index
};
struct EntryB {          // read only by pairs that share a further locus
    uint32_t next_b0;    // low bit of the base at locus + 1 + t
    uint32_t next_b1;    // high bit of the base at locus + 1 + t
    uint32_t read;       // index of the live read (segment) in read_off
    uint32_t pad;
};

constexpr uint32_t kMetaBaseShift = 16;
constexpr uint32_t kMetaTail = 1u << 18;
constexpr uint32_t kMetaPrevOvf = 1u << 19;
constexpr uint32_t kMetaNextOvf = 1u << 20;
constexpr uint32_t kWindow = 32;

struct PackedPileup {
    uint32_t num_cells = 0;
    uint32_t block_cells = 0;   // B: cells per tile edge
    uint32_t num_blocks = 0;    // ceil(num_cells / B)
    uint32_t num_loci = 0;
    uint64_t num_entries = 0;   // kept entries
    uint64_t num_reads = 0;     // live reads (segments)
    uint64_t raw_entries = 0;   // entries of the input pileup
    uint64_t pair_bound = 0;    // max over cells of sum_l n_cell(l)^2 (Cauchy-Schwarz bound)
    bool any_window_overflow = false;

    // entries sorted by (cell block, locus, input order); blk_off[b * (L+1) + l] is the first
    // entry of block b at locus l, blk_off[b * (L+1) + L] the end of block b
    std::vector<uint32_t> blk_off;
    std::vector<EntryA> entry_a;
    std::vector<EntryB> entry_b;
    // per live read: its kept entries in locus order (slow path: windows overflowed)
    std::vector<uint32_t> read_off;   // num_reads + 1
    std::vector<uint32_t> read_locus; // num_entries
    std::vector<uint8_t> read_base;   // num_entries
};

// Returns an empty string on success, else an error message (invalid input).
std::string pack_pileup(const FlatPileupView &in, uint32_t num_cells, uint32_t max_fragment_length,
                        uint32_t num_threads, uint32_t block_cells, PackedPileup *out);

}  // namespace secedo
