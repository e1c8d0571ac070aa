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

#include <cstdint>
#include <stdexcept>
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

    // flatten vector<vector<PosData>> into the structure-of-arrays pileup of secedo_simmat.h
    std::vector<uint32_t> chr_locus_off(1, 0), locus_pos, read_ids;
    std::vector<uint64_t> locus_entry_off(1, 0);
    std::vector<uint16_t> id_base;
    uint64_t n_loci = 0, n_entries = 0;
    for (const auto &chromosome : pos_data) {
        n_loci += chromosome.size();
        for (const PosDataT &pd : chromosome) n_entries += pd.read_ids.size();
    }
    locus_pos.reserve(n_loci);
    locus_entry_off.reserve(n_loci + 1);
    read_ids.reserve(n_entries);
    id_base.reserve(n_entries);
    for (const auto &chromosome : pos_data) {
        for (const PosDataT &pd : chromosome) {
            locus_pos.push_back(pd.position);
            read_ids.insert(read_ids.end(), pd.read_ids.begin(), pd.read_ids.end());
            id_base.insert(id_base.end(), pd.group_ids_bases.begin(), pd.group_ids_bases.end());
            locus_entry_off.push_back(read_ids.size());
        }
        chr_locus_off.push_back(static_cast<uint32_t>(locus_pos.size()));
    }

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
            chr_locus_off.data(), static_cast<uint32_t>(pos_data.size()), locus_pos.data(),
            locus_entry_off.data(), read_ids.data(), id_base.data(), nullptr, group_id_to_pos.data(),
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
