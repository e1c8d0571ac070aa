/*
 * em_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's expectation-maximisation refinement of a two-way
 * split (reference: expectation_maximization.cpp:19-161; caller spectral_clustering.cpp:375-377).
 * Checker for secedo_em_refine* of secedo_amd. Pinned against the
 * compiled reference (oracle/_ref, ref_em) on the five cases of the reference's own
 * tests/test_expectation_maximization.cpp and on random pileups (tests/golden/em_*.npz).
 *
 * Two things the reference does that a reader may not expect are kept, because they shape the result:
 *  - the cluster centres weight an entry by prob[group id] (expectation_maximization.cpp:24), while
 *    the likelihoods are accumulated at id_to_pos[group id] (:78-79);
 *  - the per-cell log-likelihood sums are NOT reset between iterations (:130-147: declared outside
 *    the loop, += inside it).
 */
#ifndef EM_ORACLE_H
#define EM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Flat pileup layout of simmat_oracle.h (positions and read ids are not used by this step).
 * prob_cluster_b[n_cells] in/out. Returns the number of iterations run (>= 1), or
 *   -1 a group id >= n_cells would index prob_cluster_b out of bounds (the reference reads past the
 *      vector there), -2 a group id >= n_groups or id_to_pos[group] >= n_cells (the reference's
 *      vector::at throws), -3 more than max_iterations iterations.
 */
int oracle_em(const uint32_t *chr_locus_off, uint32_t n_chr, const uint64_t *locus_entry_off,
              const uint32_t *id_base, const uint32_t *id_to_pos, uint32_t n_groups, double theta,
              double *prob_cluster_b, uint32_t n_cells, uint32_t max_iterations);

#ifdef __cplusplus
}
#endif

#endif
