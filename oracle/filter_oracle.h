/*
 * filter_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's locus filter, the step immediately upstream of the
 * similarity matrix (reference: util/is_significant.cpp:78-138 Filter::is_significant, :149-193
 * Filter::filter; caller spectral_clustering.cpp:336-337). Checker for the HIP filter of
 * secedo_amd. Pinned against the compiled reference
 * (oracle/_ref, ref_filter / ref_is_significant) and against tests/golden/filter_*.npz and the
 * known-answer strings of the reference's tests/test_is_significant.cpp:46-90.
 */
#ifndef FILTER_ORACLE_H
#define FILTER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Marks a group id outside the current sub-cluster in id_to_pos (util/is_significant.hpp:11). */
#define ORACLE_NO_POS 16383u

/* Filter::is_significant(base_count) for counts of A, C, G, T (util/is_significant.cpp:78-138).
 * cell_proportion 0..4 selects the threshold row (util/is_significant.hpp:30-40). Returns 0/1. */
int oracle_is_significant(const uint16_t base_count[4], double theta, uint32_t cell_proportion);

/* log P(counts | homozygous) - log evidence and the threshold it is compared with (for margin checks) */
void oracle_significance_terms(const uint16_t base_count[4], double theta, uint32_t cell_proportion,
                               double *statistic, double *threshold);

/*
 * Filter::filter (util/is_significant.cpp:149-193) on the flat pileup layout of simmat_oracle.h:
 * entries whose group g has id_to_pos[g] == ORACLE_NO_POS are dropped, a locus is kept iff the base
 * counts of the remaining entries are significant. Outputs are caller-allocated with the input's
 * capacities; out_chr_locus_off has n_chr + 1 entries. Returns 0.
 */
int oracle_filter(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
                  const uint64_t *locus_entry_off, const uint32_t *read_ids, const uint32_t *id_base,
                  const uint32_t *id_to_pos, uint32_t n_groups, double theta, uint32_t cell_proportion,
                  uint32_t *out_chr_locus_off, uint32_t *out_locus_pos, uint64_t *out_locus_entry_off,
                  uint32_t *out_read_ids, uint32_t *out_id_base, uint64_t *out_n_loci,
                  uint64_t *out_n_entries, double *avg_coverage);

#ifdef __cplusplus
}
#endif
#endif
