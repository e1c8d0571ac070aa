// filter_device.hpp -- device-side locus filter (see filter_device.hip).
#pragma once

#include "pack_device.hpp"

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

namespace secedo {

struct FilterWorkspace {
    DeviceArena a, b;
};

// device output arrays, caller-allocated with the input's capacities (n_chr + 1, n_loci,
// n_loci + 1, n_entries, n_entries)
struct FilterOut {
    uint32_t *chr_locus_off;
    uint32_t *locus_pos;
    uint64_t *locus_entry_off;
    uint32_t *read_ids;
    void *id_base;  // same width as the input's
};

// in.group_id_to_pos is the reference's id_to_pos (kNoPos marks groups outside the sub-cluster).
// Returns "" on success. Synchronises `stream` (two scalar read-backs).
std::string filter_device(const DeviceFlatPileup &in, double theta, uint32_t cell_proportion,
                          hipStream_t stream, FilterWorkspace *ws, const FilterOut &out, uint64_t *n_loci_out,
                          uint64_t *n_entries_out, double *avg_coverage);

}  // namespace secedo
