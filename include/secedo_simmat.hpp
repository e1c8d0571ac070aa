// secedo_simmat.hpp -- header-only C++17 shim that keeps the reference's C++ signature
//
//     Matd computeSimilarityMatrix(const std::vector<std::vector<PosData>> &pos_data,
//                                  uint32_t num_cells, uint32_t max_fragment_length,
//                                  const std::vector<uint32_t> &group_id_to_pos,
//                                  double mutation_rate, double homozygous_rate,
//                                  double seq_error_rate, const uint32_t num_threads,
//                                  const std::string &marker, const std::string &normalization);
//
// (reference: similarity_matrix.hpp:51-60) on top of the C-ABI of secedo_simmat.h. It is a template
// over the matrix and the per-locus record type so that it compiles unchanged against the
// reference's own `Matd` (util/mat.hpp) and `PosData` (sequenced_data.hpp) -- see INTEGRATION.md --
// and against the stand-ins of this repository's C++ test (tests/cpp/).
//
// Requirements on the types (all met by the reference's):
//   PosDataT: members `uint32_t position`, `std::vector<uint32_t> read_ids`,
//             `std::vector<uint16_t> group_ids_bases`           (sequenced_data.hpp:26-37)
//   MatdT:    constructor MatdT(rows, cols) owning its elements, `double &operator()(r, c)`
//             (util/mat.hpp:86, :117). When MatdT also has `double *data()` over contiguous row-major
//             storage -- the reference's Mat<double> does, util/mat.hpp:238 with :117 -- the library
//             writes the result straight into it; otherwise it goes through a buffer and operator().
//
// Error behaviour of the reference is kept: an unknown normalisation throws std::logic_error
// ("Invalid normalization: ..."), here before any work is done (the reference throws after the
// accumulation, similarity_matrix.cpp:264 via :430; no result is produced either way). Any other
// failure of the library (no GPU, invalid pileup) throws std::runtime_error with the library's
// message; the reference has no error path there (it asserts or exits).
#pragma once

#include "secedo_simmat.h"

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <thread>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace secedo_amd {

namespace detail {
// does `m.data()` give a double* (contiguous storage the library may write the matrix into)?
template <class M, class = void>
struct has_double_data : std::false_type {};
template <class M>
struct has_double_data<M, std::void_t<decltype(std::declval<M &>().data())>>
    : std::is_convertible<decltype(std::declval<M &>().data()), double *> {};
}  // namespace detail

template <class MatdT, class PosDataT>
MatdT computeSimilarityMatrix(const std::vector<std::vector<PosDataT>> &pos_data, uint32_t num_cells,
                              uint32_t max_fragment_length,
                              const std::vector<uint32_t> &group_id_to_pos, double mutation_rate,
                              double homozygous_rate, double seq_error_rate,
                              const uint32_t num_threads, const std::string & /*marker*/,
                              const std::string &normalization) {
    const int norm = secedo_simmat_normalization_from_string(normalization.c_str());
    if (norm < 0) throw std::logic_error("Invalid normalization: " + normalization);

    // Flatten vector<vector<PosData>> into the structure-of-arrays pileup of secedo_simmat.h: straight into the
    // library's page-locked staging buffers (no fresh pages per call, uploaded by DMA), the loci shared out
    // among up to num_threads threads (the reference spends its num_threads on this call too). Buffers of our
    // own when the staging is taken (a concurrent caller) or unavailable.
    const size_t n_chr = pos_data.size();
    std::vector<const PosDataT *> loci;
    {
        size_t n = 0;
        for (const auto &chromosome : pos_data) n += chromosome.size();
        loci.reserve(n);
    }
    std::vector<uint32_t> chr_off_own(1, 0);
    for (const auto &chromosome : pos_data) {
        for (const PosDataT &pd : chromosome) loci.push_back(&pd);
        chr_off_own.push_back(static_cast<uint32_t>(loci.size()));
    }
    const size_t n_loci = loci.size();
    unsigned n_workers = std::max(1u, std::min({num_threads, std::thread::hardware_concurrency(), 16u}));
    if (n_loci < 4096) n_workers = 1;
    std::vector<uint64_t> first_entry(n_workers + 1, 0);  // entries before each worker's loci
    auto share = [&](unsigned t) { return n_loci * t / n_workers; };
    auto run = [&](auto &&body) {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_workers; ++t) pool.emplace_back(body, t);
        body(0u);
        for (auto &th : pool) th.join();
    };
    run([&](unsigned t) {
        uint64_t n = 0;
        for (size_t l = share(t); l < share(t + 1); ++l) n += loci[l]->read_ids.size();
        first_entry[t + 1] = n;
    });
    for (unsigned t = 0; t < n_workers; ++t) first_entry[t + 1] += first_entry[t];
    const uint64_t n_entries = first_entry[n_workers];

    std::vector<uint32_t> locus_pos_own, read_ids_own;
    std::vector<uint64_t> locus_entry_off_own;
    std::vector<uint16_t> id_base_own;
    uint32_t *chr_locus_off, *locus_pos, *read_ids;
    uint64_t *locus_entry_off;
    uint16_t *id_base;
    const uint64_t want[5] = {(n_chr + 1) * 4, n_loci * 4 + 4, (n_loci + 1) * 8, n_entries * 4 + 4, n_entries * 2 + 4};
    void *staged[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    struct StagingGuard {
        bool held = false;
        ~StagingGuard() { if (held) secedo_simmat_staging_release(); }
    } staging_guard;
    if (secedo_simmat_staging_acquire(want, staged) == SECEDO_OK) {
        staging_guard.held = true;
        chr_locus_off = static_cast<uint32_t *>(staged[0]);
        locus_pos = static_cast<uint32_t *>(staged[1]);
        locus_entry_off = static_cast<uint64_t *>(staged[2]);
        read_ids = static_cast<uint32_t *>(staged[3]);
        id_base = static_cast<uint16_t *>(staged[4]);
        std::copy(chr_off_own.begin(), chr_off_own.end(), chr_locus_off);
    } else {
        locus_pos_own.resize(n_loci + 1);
        locus_entry_off_own.resize(n_loci + 1);
        read_ids_own.resize(n_entries + 1);
        id_base_own.resize(n_entries + 1);
        chr_locus_off = chr_off_own.data();
        locus_pos = locus_pos_own.data();
        locus_entry_off = locus_entry_off_own.data();
        read_ids = read_ids_own.data();
        id_base = id_base_own.data();
    }
    run([&](unsigned t) {
        uint64_t e = first_entry[t];
        for (size_t l = share(t); l < share(t + 1); ++l) {
            const PosDataT &pd = *loci[l];
            locus_pos[l] = pd.position;
            locus_entry_off[l] = e;
            std::copy(pd.read_ids.begin(), pd.read_ids.end(), read_ids + e);
            std::copy(pd.group_ids_bases.begin(), pd.group_ids_bases.end(), id_base + e);
            e += pd.read_ids.size();
        }
    });
    locus_entry_off[n_loci] = n_entries;

    MatdT result(num_cells, num_cells);
    std::vector<double> staging;  // only for matrix types without contiguous data()
    double *out;
    if constexpr (detail::has_double_data<MatdT>::value) {
        out = result.data();
    } else {
        staging.resize(static_cast<size_t>(num_cells) * num_cells);
        out = staging.data();
    }
    const int rc = secedo_simmat_compute(
            chr_locus_off, static_cast<uint32_t>(n_chr), locus_pos, locus_entry_off, read_ids, id_base, nullptr,
            group_id_to_pos.data(),
            static_cast<uint32_t>(group_id_to_pos.size()), num_cells, max_fragment_length,
            mutation_rate, homozygous_rate, seq_error_rate, num_threads, norm, out);
    if (rc == SECEDO_E_INVALID_NORMALIZATION) throw std::logic_error("Invalid normalization: " + normalization);
    if (rc != SECEDO_OK) throw std::runtime_error(std::string("secedo_simmat: ") + secedo_simmat_last_error());
    if constexpr (!detail::has_double_data<MatdT>::value) {
        for (uint32_t i = 0; i < num_cells; ++i) {
            for (uint32_t j = 0; j < num_cells; ++j) result(i, j) = staging[static_cast<size_t>(i) * num_cells + j];
        }
    }
    return result;
}

}  // namespace secedo_amd
