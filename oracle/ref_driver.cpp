/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C-ABI wrapper around the UNMODIFIED reference, compiled from the sources where they lie
 * under /root/reference by oracle/Makefile into oracle/_ref/libsecedo_ref.so (git-ignored;
 * nothing from the reference is copied into this repository). It exists to
 *   (a) validate oracle/simmat_oracle.c,
 *   (b) generate the vectors under tests/golden/ (oracle/gen_golden.py),
 *   (c) serve as bench.py's cpu_baseline of kind "reference".
 * The flat layout is the one of oracle/simmat_oracle.h.
 */
#include "expectation_maximization.hpp" // reference: expectation_maximization
#include "similarity_matrix.hpp"      // reference: computeSimilarityMatrix
#include "util/is_significant.hpp"    // reference: Filter
#include "util/pileup_reader.hpp"     // reference: read_pileup, get_grouping

#include <array>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace {

// The reference draws a progress bar on std::cout (similarity_matrix.cpp:338-341); keep the
// test output clean by parking cout on a string buffer for the duration of a call.
struct QuietCout {
    std::ostringstream sink;
    std::streambuf *old;
    QuietCout() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~QuietCout() { std::cout.rdbuf(old); }
};

thread_local std::vector<PosData> g_read_result;
thread_local uint32_t g_read_num_cells = 0, g_read_max_len = 0;

} // namespace

extern "C" {

// Calls the reference's computeSimilarityMatrix (similarity_matrix.hpp:51-60).
// id_base carries group_id<<2|base widened to u32 (ids must fit the reference's 14 bits).
// Returns 0, -2 for an invalid normalisation string (std::logic_error in the reference),
// -3 for a cell id the reference's u16 packing cannot hold, -4 for any other exception.
int ref_simmat_compute(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
                       const uint64_t *locus_entry_off, const uint32_t *read_ids,
                       const uint32_t *id_base, const uint32_t *group_id_to_pos,
                       uint32_t n_groups, uint32_t num_cells, uint32_t max_fragment_length,
                       double mutation_rate, double homozygous_rate, double seq_error_rate,
                       uint32_t num_threads, const char *normalization, double *out) {
    std::vector<std::vector<PosData>> pos_data(n_chr);
    for (uint32_t c = 0; c < n_chr; ++c) {
        for (uint32_t l = chr_locus_off[c]; l < chr_locus_off[c + 1]; ++l) {
            const uint64_t b = locus_entry_off[l], e = locus_entry_off[l + 1];
            std::vector<uint32_t> ids(read_ids + b, read_ids + e);
            std::vector<uint16_t> packed(e - b);
            for (uint64_t i = b; i < e; ++i) {
                if (id_base[i] > 0xFFFFu) return -3;
                packed[i - b] = static_cast<uint16_t>(id_base[i]);
            }
            pos_data[c].emplace_back(locus_pos[l], std::move(ids), std::move(packed));
        }
    }
    std::vector<uint32_t> g2p(group_id_to_pos, group_id_to_pos + n_groups);
    try {
        QuietCout quiet;
        Matd m = computeSimilarityMatrix(pos_data, num_cells, max_fragment_length, g2p,
                                         mutation_rate, homozygous_rate, seq_error_rate,
                                         num_threads, "", normalization);
        if (m.rows() != num_cells || m.cols() != num_cells) return -4;
        for (uint32_t i = 0; i < num_cells; ++i) {
            for (uint32_t j = 0; j < num_cells; ++j) {
                out[static_cast<uint64_t>(i) * num_cells + j] = m(i, j);
            }
        }
    } catch (const std::logic_error &) {
        return -2;
    } catch (...) {
        return -4;
    }
    return 0;
}

// Runs the reference's pileup reader (util/pileup_reader.hpp:33-39) on `fname` (text or .bin)
// with the grouping get_grouping(merge_count, merge_file) would produce, keeps the result in
// thread-local storage and reports its size; ref_read_pileup_fetch copies it out.
// NOTE: the reference's text reader writes `fname + ".bin"` next to its input
// (util/pileup_reader.cpp:20), so callers must pass a copy that lives in a scratch directory.
int ref_read_pileup(const char *fname, uint32_t merge_count, const char *merge_file,
                    uint32_t max_coverage, uint64_t *n_loci, uint64_t *n_entries,
                    uint32_t *num_cells, uint32_t *max_len) {
    try {
        std::vector<uint16_t> id_to_group
                = get_grouping(static_cast<uint16_t>(merge_count), merge_file ? merge_file : "");
        auto [pds, cells, len]
                = read_pileup(fname, id_to_group, [](uint64_t) {}, max_coverage, {}, true);
        g_read_result = std::move(pds);
        g_read_num_cells = cells;
        g_read_max_len = len;
    } catch (...) {
        return -4;
    }
    uint64_t e = 0;
    for (const PosData &pd : g_read_result) e += pd.size();
    *n_loci = g_read_result.size();
    *n_entries = e;
    *num_cells = g_read_num_cells;
    *max_len = g_read_max_len;
    return 0;
}

void ref_read_pileup_fetch(uint32_t *locus_pos, uint64_t *locus_entry_off, uint32_t *read_ids,
                           uint32_t *id_base) {
    uint64_t e = 0;
    uint64_t l = 0;
    for (const PosData &pd : g_read_result) {
        locus_pos[l] = pd.position;
        locus_entry_off[l] = e;
        for (uint32_t i = 0; i < pd.size(); ++i, ++e) {
            read_ids[e] = pd.read_ids[i];
            id_base[e] = pd.group_ids_bases[i];
        }
        ++l;
    }
    locus_entry_off[l] = e;
    g_read_result.clear();
}

// expectation_maximization (expectation_maximization.cpp:125-161) on the flat layout. prob_cluster_b
// in/out. Returns 0, -3 for an id the reference's u16 packing cannot hold, -4 for an exception
// (vector::at on a group outside id_to_pos / a position outside the probability vector).
int ref_em(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
           const uint64_t *locus_entry_off, const uint32_t *read_ids, const uint32_t *id_base,
           const uint32_t *id_to_pos, uint32_t n_groups, double theta, double *prob_cluster_b, uint32_t n_cells) {
    std::vector<std::vector<PosData>> pos_data(n_chr);
    for (uint32_t c = 0; c < n_chr; ++c) {
        for (uint32_t l = chr_locus_off[c]; l < chr_locus_off[c + 1]; ++l) {
            const uint64_t b = locus_entry_off[l], e = locus_entry_off[l + 1];
            std::vector<uint32_t> ids(read_ids + b, read_ids + e);
            std::vector<uint16_t> packed(e - b);
            for (uint64_t i = b; i < e; ++i) {
                if (id_base[i] > 0xFFFFu) return -3;
                packed[i - b] = static_cast<uint16_t>(id_base[i]);
            }
            pos_data[c].emplace_back(locus_pos[l], std::move(ids), std::move(packed));
        }
    }
    std::vector<uint32_t> i2p(id_to_pos, id_to_pos + n_groups);
    std::vector<double> prob(prob_cluster_b, prob_cluster_b + n_cells);
    try {
        QuietCout quiet;
        expectation_maximization(pos_data, i2p, 1, theta, &prob);
    } catch (...) {
        return -4;
    }
    std::memcpy(prob_cluster_b, prob.data(), n_cells * sizeof(double));
    return 0;
}

// Filter::is_significant on base counts (util/is_significant.cpp:78-138).
int ref_is_significant(const uint16_t *base_count, double theta, uint32_t cell_proportion) {
    Filter filter(theta, static_cast<uint8_t>(cell_proportion));
    std::array<uint16_t, 4> counts = { base_count[0], base_count[1], base_count[2], base_count[3] };
    return filter.is_significant(counts) ? 1 : 0;
}

// Filter::filter (util/is_significant.cpp:149-193) on the flat layout; outputs caller-allocated with
// the input's capacities. ids must fit the reference's 14 bits.
int ref_filter(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
               const uint64_t *locus_entry_off, const uint32_t *read_ids, const uint32_t *id_base,
               const uint32_t *id_to_pos, uint32_t n_groups, double theta, uint32_t cell_proportion,
               uint32_t num_threads, uint32_t *out_chr_locus_off, uint32_t *out_locus_pos,
               uint64_t *out_locus_entry_off, uint32_t *out_read_ids, uint32_t *out_id_base,
               uint64_t *out_n_loci, uint64_t *out_n_entries, double *avg_coverage) {
    std::vector<std::vector<PosData>> pos_data(n_chr);
    for (uint32_t c = 0; c < n_chr; ++c) {
        for (uint32_t l = chr_locus_off[c]; l < chr_locus_off[c + 1]; ++l) {
            const uint64_t b = locus_entry_off[l], e = locus_entry_off[l + 1];
            std::vector<uint32_t> ids(read_ids + b, read_ids + e);
            std::vector<uint16_t> packed(e - b);
            for (uint64_t i = b; i < e; ++i) {
                if (id_base[i] > 0xFFFFu) return -3;
                packed[i - b] = static_cast<uint16_t>(id_base[i]);
            }
            pos_data[c].emplace_back(locus_pos[l], std::move(ids), std::move(packed));
        }
    }
    std::vector<uint32_t> i2p(id_to_pos, id_to_pos + n_groups);
    try {
        Filter filter(theta, static_cast<uint8_t>(cell_proportion));
        auto [result, cov] = filter.filter(pos_data, i2p, "", num_threads);
        uint64_t nl = 0, ne = 0;
        out_chr_locus_off[0] = 0;
        out_locus_entry_off[0] = 0;
        for (uint32_t c = 0; c < n_chr; ++c) {
            for (const PosData &pd : result[c]) {
                out_locus_pos[nl] = pd.position;
                for (uint32_t i = 0; i < pd.size(); ++i, ++ne) {
                    out_read_ids[ne] = pd.read_ids[i];
                    out_id_base[ne] = pd.group_ids_bases[i];
                }
                ++nl;
                out_locus_entry_off[nl] = ne;
            }
            out_chr_locus_off[c + 1] = static_cast<uint32_t>(nl);
        }
        *out_n_loci = nl;
        *out_n_entries = ne;
        *avg_coverage = cov;
    } catch (...) {
        return -4;
    }
    return 0;
}

} // extern "C"
