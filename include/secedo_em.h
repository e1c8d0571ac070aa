/*
 * secedo_em.h -- C-ABI of the expectation-maximisation refinement of a two-way split
 * (SURVEY.md section 8f, rank 4).
 *
 * Replaces the reference's
 *     void expectation_maximization(const std::vector<std::vector<PosData>> &pos_data,
 *                                   const std::vector<uint32_t> &id_to_pos, uint32_t num_threads,
 *                                   double theta, std::vector<double> *prob_cluster_b)
 * (expectation_maximization.hpp:26-30, implementation expectation_maximization.cpp:19-161, sole
 * caller spectral_clustering.cpp:375-377, on the filtered pileup of the sub-cluster).
 * num_threads is unused by the reference (:127) and has no counterpart here.
 *
 * Two behaviours of the reference are part of the contract because they shape the result:
 *   - a cluster centre weights an entry by prob_cluster_b[group id] (:24), while the per-cell
 *     likelihoods are accumulated at id_to_pos[group id] (:78-79). A group id >= n_cells makes the
 *     reference read past the vector; here that is SECEDO_E_INVALID_ARG.
 *   - the per-cell log-likelihood sums are not reset between iterations (:130-147).
 * The reference iterates until no probability moves by 1e-2 (:105, :118) without a bound; here
 * max_iterations (0 -> 1000) bounds the loop and exceeding it is SECEDO_E_LIMIT.
 *
 * Flat pileup layout of secedo_simmat.h (positions and read ids are not used by this step). Error
 * codes and secedo_simmat_last_error() are those of secedo_simmat.h. No CPU fallback.
 */
#ifndef SECEDO_EM_H
#define SECEDO_EM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Pileup and probabilities resident in HBM (e.g. the output of secedo_filter_device).
 * d_prob_cluster_b[n_cells] in/out; *iterations (may be NULL) = maximisation/expectation rounds run. */
int secedo_em_refine_device(int device_id, const uint64_t *d_locus_entry_off, uint32_t n_loci,
                            uint64_t n_entries, const uint16_t *d_id_base16, const uint32_t *d_id_base32,
                            const uint32_t *d_id_to_pos, uint32_t n_groups, double theta,
                            double *d_prob_cluster_b, uint32_t n_cells, uint32_t max_iterations,
                            uint32_t *iterations, void *stream);

/* Host buffers in, host probabilities out: what a caller holding a std::vector<double> uses. */
int secedo_em_refine(int device_id, const uint64_t *locus_entry_off, uint32_t n_loci,
                     const uint16_t *id_base16, const uint32_t *id_base32, const uint32_t *id_to_pos,
                     uint32_t n_groups, double theta, double *prob_cluster_b, uint32_t n_cells,
                     uint32_t max_iterations, uint32_t *iterations);

#ifdef __cplusplus
}
#endif

#endif /* SECEDO_EM_H */
