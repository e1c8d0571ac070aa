// secedo_pipeline.hpp -- header-only C++ host side of the two consumers next to the similarity matrix,
// with the reference's own signatures on top of the C-ABI (secedo_em.h, secedo_spectral.h). Templates
// over the caller's types, like secedo_simmat.hpp, so that this repository contains none of the
// reference's headers.
//
//   void expectation_maximization(pos_data, id_to_pos, num_threads, theta, &prob_cluster_b)
//        reference: expectation_maximization.hpp:26-30 (caller spectral_clustering.cpp:375-377)
//   void smallest_eigenpairs(similarity, n_values, n_vectors, &eigenvalues, &eigenvectors)
//        replaces laplacian() + arma::eig_sym in spectral_clustering() (spectral_clustering.cpp:127-138);
//        eigenvectors column-major n x n_vectors (arma::mat layout)
//
// A failure of the library throws std::runtime_error with the library's message; the reference has no
// error path at these places (it asserts, or reads out of bounds).
#pragma once

#include "secedo_em.h"
#include "secedo_simmat.h"
#include "secedo_spectral.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace secedo_amd {

// vector<vector<PosData>> -> the structure-of-arrays pileup of secedo_simmat.h
struct FlatPileupHost {
    std::vector<uint32_t> chr_locus_off{0}, locus_pos, read_ids;
    std::vector<uint64_t> locus_entry_off{0};
    std::vector<uint16_t> id_base;
    uint32_t n_loci() const { return static_cast<uint32_t>(locus_pos.size()); }
};

template <class PosDataT>
FlatPileupHost flatten(const std::vector<std::vector<PosDataT>> &pos_data) {
    FlatPileupHost flat;
    uint64_t n_loci = 0, n_entries = 0;
    for (const auto &chromosome : pos_data) {
        n_loci += chromosome.size();
        for (const PosDataT &pd : chromosome) n_entries += pd.read_ids.size();
    }
    flat.locus_pos.reserve(n_loci);
    flat.locus_entry_off.reserve(n_loci + 1);
    flat.read_ids.reserve(n_entries);
    flat.id_base.reserve(n_entries);
    for (const auto &chromosome : pos_data) {
        for (const PosDataT &pd : chromosome) {
            flat.locus_pos.push_back(pd.position);
            flat.read_ids.insert(flat.read_ids.end(), pd.read_ids.begin(), pd.read_ids.end());
            flat.id_base.insert(flat.id_base.end(), pd.group_ids_bases.begin(), pd.group_ids_bases.end());
            flat.locus_entry_off.push_back(flat.read_ids.size());
        }
        flat.chr_locus_off.push_back(static_cast<uint32_t>(flat.locus_pos.size()));
    }
    return flat;
}

template <class PosDataT>
void expectation_maximization(const std::vector<std::vector<PosDataT>> &pos_data,
                              const std::vector<uint32_t> &id_to_pos, uint32_t /*num_threads*/, double theta,
                              std::vector<double> *prob_cluster_b) {
    const FlatPileupHost flat = flatten(pos_data);
    const int rc = secedo_em_refine(0, flat.locus_entry_off.data(), flat.n_loci(), flat.id_base.data(), nullptr,
                                    id_to_pos.data(), static_cast<uint32_t>(id_to_pos.size()), theta,
                                    prob_cluster_b->data(), static_cast<uint32_t>(prob_cluster_b->size()), 0,
                                    nullptr);
    if (rc != SECEDO_OK) throw std::runtime_error(std::string("secedo_em: ") + secedo_simmat_last_error());
}

// MatdT: rows(), and contiguous row-major storage reachable through `data()` (the reference's
// Mat<double>) -- the matrix is symmetric, so row-major and column-major coincide.
// The iteration stops at residuals <= 1e-9 or after its cycle limit; pairs that did not get there are not
// handed out silently: without `info_out` an unconverged solve throws, with it the caller decides
// (info_out->converged, max_residual_vectors / _values).
template <class MatdT>
void smallest_eigenpairs(const MatdT &similarity, uint32_t n_values, uint32_t n_vectors,
                         std::vector<double> *eigenvalues, std::vector<double> *eigenvectors,
                         secedo_spectral_info *info_out = nullptr) {
    const uint32_t n = similarity.rows();
    n_values = n_values < n ? n_values : n;
    n_vectors = n_vectors < n_values ? n_vectors : n_values;
    eigenvalues->assign(n_values, 0.0);
    eigenvectors->assign(static_cast<size_t>(n) * n_vectors, 0.0);
    secedo_spectral_info info;
    const int rc = secedo_spectral_eigs(0, similarity.data(), n, n_values, n_vectors, 0.0, 0, eigenvalues->data(),
                                        eigenvectors->data(), &info);
    if (rc != SECEDO_OK) throw std::runtime_error(std::string("secedo_spectral: ") + secedo_simmat_last_error());
    if (info_out) *info_out = info;
    else if (!info.converged)
        throw std::runtime_error("secedo_spectral: the eigensolver did not converge (residual "
                                 + std::to_string(info.max_residual_vectors) + " after "
                                 + std::to_string(info.cycles) + " cycles)");
}

}  // namespace secedo_amd
