// pack_device.hpp -- device-side read assembly, flush schedule and tile packing (gfx950).
//
// Same result as pack_host.cpp (see its header for the reference semantics) but computed on the
// GPU from the raw flat pileup resident in HBM. Reads that outlive max_fragment_length are cut where
// a flush erases them (the cuts and the flush chain are iterated to their fixed point). When a size
// limit of this path is exceeded (positions within max_fragment_length of 2^32, 2^31 entries, an
// empty pileup) or that iteration does not settle, `need_host` is set and the caller packs on the host.
#pragma once

#include "pack_host.hpp"

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

namespace secedo {

// Raw flat pileup, device pointers (layout of include/secedo_simmat.h)
struct DeviceFlatPileup {
    const uint32_t *chr_locus_off = nullptr;
    uint32_t n_chr = 0;
    const uint32_t *locus_pos = nullptr;
    const uint64_t *locus_entry_off = nullptr;
    const uint32_t *read_ids = nullptr;
    const uint16_t *id_base16 = nullptr;
    const uint32_t *id_base32 = nullptr;
    const uint32_t *group_id_to_pos = nullptr;
    uint32_t n_groups = 0;
    uint32_t n_loci = 0;
    uint64_t n_entries = 0;
};

// Growable device allocation owned by the caller (simmat_api.cpp's handle)
int poison_level();  // SECEDO_POISON, a debugging aid (pack_device.hip)

struct DeviceArena {
    void *p = nullptr;
    size_t bytes = 0;
    ~DeviceArena();
    hipError_t ensure(size_t n);
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

// The packed pileup in HBM: the arrays of PackedPileup, device resident
struct DevicePacked {
    DeviceArena blk_off, entry32, mask32, entry, entry_read, range_off, read_off, read_locus, read_base;
    DeviceArena scratch[24];  // temporaries, kept between calls to avoid re-allocation
    uint32_t num_cells = 0, block_cells = 0, num_blocks = 0, num_loci = 0, num_ranges = 0;
    uint64_t num_entries = 0, num_reads = 0, pair_bound = 0, multi_entries = 0;
    uint32_t max_read_entries = 0;  // kept entries of the longest read (before flush splits: an upper bound)
    uint32_t n_wide = 0;            // kept entries whose read reaches beyond their 8-locus windows (C_WIDE in entry32)
    // per matrix row: sum over loci of (kept entries of the row at the locus)^2 -- pair_bound is its maximum.
    // Device packing: a pointer into the packing's scratch, valid until the next packing of this handle;
    // host packing: cell_sq_host. secedo_simmat_cell_squares() hands it out (shards that are added up
    // sum these vectors to get the exact bound of the union).
    const unsigned long long *cell_sq = nullptr;
    uint32_t cell_sq_n = 0;
    std::vector<uint64_t> cell_sq_host;
    bool stage_masks = false;
    bool count_tile = false;
    uint32_t cap_entries = 0, cap_loci = 0;
    uint64_t id_space_hint = 0;  // size of the read-id space of the previous call (0: unknown): saves a read-back
    // the single-entry fast path pays when most reads have one entry: a call that finds more than half of the
    // entries in multi-entry reads finishes without it, and the handle's next calls do not try (every 16th does)
    bool split_pays = true;
    uint32_t calls_without_split = 0;
    // side stream for the branch of the pipeline nothing else waits for until the final gather
    // (completed counts + flush chain); created on first use
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_offsets = nullptr, ev_flush = nullptr;
    // pinned host words the packing's scalar read-backs arrive in (polled; see read_scalars)
    void *mailbox = nullptr;
    unsigned long long mailbox_seq = 0;
    bool mailbox_failed = false;
    DevicePacked() = default;
    DevicePacked(const DevicePacked &) = delete;
    DevicePacked &operator=(const DevicePacked &) = delete;
    ~DevicePacked();
};

// Returns "" on success. On success with *need_host == true nothing usable was produced and the
// caller must fall back to pack_pileup() on the host. Synchronises `stream` a few times (scalar
// read-backs).
std::string pack_pileup_device(const DeviceFlatPileup &in, uint32_t num_cells,
                               uint32_t max_fragment_length, uint32_t num_threads,
                               uint32_t block_cells, StageGeometry (*geometry)(uint32_t),
                               bool allow_count_tile, hipStream_t stream, DevicePacked *out,
                               bool *need_host);

}  // namespace secedo
