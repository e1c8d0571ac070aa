/*
 * simmat_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the algorithm of the reference's
 * computeSimilarityMatrix() (reference: similarity_matrix.cpp:295-433). It is the
 * checker for the HIP path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it. The product (secedo_amd/) never links or calls it.
 *
 * Parity pin: validated against the *compiled, unmodified* reference
 * (oracle/_ref/libsecedo_ref.so, built by oracle/Makefile from the sources where
 * they lie under /root/reference) and against the committed vectors in
 * tests/golden/ that were generated from that build (oracle/gen_golden.py).
 * The reference's own test-suite holds no numeric vector for this path
 * (SURVEY.md section 8c).
 */
#ifndef SIMMAT_ORACLE_H
#define SIMMAT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_ADD_MIN = 0, ORACLE_EXPONENTIATE = 1, ORACLE_SCALE_MAX_1 = 2 };

/*
 * Flat pileup layout (the same one the product's C-ABI takes, include/secedo_simmat.h):
 *   chr_locus_off[n_chr+1]  loci of chromosome c are [chr_locus_off[c], chr_locus_off[c+1])
 *   locus_pos[L]            PosData::position            (sequenced_data.hpp:26)
 *   locus_entry_off[L+1]    entries of locus l are [off[l], off[l+1])
 *   read_ids[E]             PosData::read_ids            (sequenced_data.hpp:28)
 *   id_base[E]              group_id<<2 | base, widened to u32 (sequenced_data.hpp:29-37)
 *
 * out      : num_cells^2 doubles, row-major, the normalised matrix the reference returns.
 * out_raw  : optional (may be NULL) num_cells^2 doubles, mat_diff - mat_same before
 *            normalisation (similarity_matrix.cpp:428).
 * Returns 0, or a negative code for an invalid argument; -3 (ORACLE_E_OUT_OF_TABLES) when a read pair
 * shares >= max_fragment_length loci, where the reference reads past its tables (:314-317, :330).
 */
int oracle_simmat_compute(const uint32_t *chr_locus_off,
                          uint32_t n_chr,
                          const uint32_t *locus_pos,
                          const uint64_t *locus_entry_off,
                          const uint32_t *read_ids,
                          const uint32_t *id_base,
                          const uint32_t *group_id_to_pos,
                          uint32_t n_groups,
                          uint32_t num_cells,
                          uint32_t max_fragment_length,
                          double mutation_rate,
                          double homozygous_rate,
                          double seq_error_rate,
                          uint32_t num_threads,
                          int normalization,
                          double *out,
                          double *out_raw);

/* log P(x_s, x_d | same genotype) and | different genotype), evaluated exactly as the
 * reference evaluates them (similarity_matrix.cpp:153-170 and :117-141), including the
 * u64 Pascal triangle (similarity_matrix.cpp:95-101). table_size plays the role of
 * max_fragment_length in Cache (similarity_matrix.cpp:76-103); x_s + x_d < table_size. */
double oracle_log_prob_same(uint32_t x_s, uint32_t x_d, double mutation_rate,
                            double homozygous_rate, double seq_error_rate, uint32_t table_size);
double oracle_log_prob_diff(uint32_t x_s, uint32_t x_d, double mutation_rate,
                            double homozygous_rate, double seq_error_rate, uint32_t table_size);

/* 0 (default): u64 binomials with the reference's wrap-around, bit-compatible with the reference.
 * 1: the same nested sums with a long-double Pascal triangle (the reference formula in exact
 * arithmetic), for inputs whose x_s + x_d exceeds the reference's own numeric range (~48).
 * Thread-local. */
void oracle_set_exact_binomials(int on);

/* 0 (default): two sums + final subtraction as in the reference (similarity_matrix.cpp:428).
 * 1: add (log P_diff - log P_same) per pair (no cancellation); see simmat_oracle.c. Thread-local. */
void oracle_set_direct_llr_sum(int on);

/* In-place normalisation of an n x n matrix (similarity_matrix.cpp:271-293). */
int oracle_normalize(int normalization, double *mat, uint32_t n);

/* Number of (read pair, shared locus) incidences counted by the last oracle_simmat_compute
 * call on this thread (the "updates" U of SURVEY.md section 8d), and the number of read pairs. */
uint64_t oracle_last_updates(void);
uint64_t oracle_last_read_pairs(void);

#ifdef __cplusplus
}
#endif
#endif
