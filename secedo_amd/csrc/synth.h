/*
 * synth.h -- SYNTH-v1 synthetic pileup generator: a bench and test utility, host
 * only. NOT part of the drop-in boundary (include/): it lives in its own library,
 * secedo_amd/libsecedo_synth.so, so that the product library exports nothing but the interfaces the
 * reference has.
 */
#ifndef SECEDO_SYNTH_H
#define SECEDO_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Call with all output pointers NULL to obtain the sizes, then again with buffers. Returns 0, or -1 for an
 * invalid spec. */
typedef struct secedo_synth_spec {
    uint32_t num_cells;
    uint32_t num_loci;
    uint32_t num_chromosomes;
    uint32_t gap_max;      /* locus gaps are 1 + rng % gap_max */
    double new_frag_prob;  /* per cell and locus probability of a new fragment (p) */
    uint32_t frag_min, frag_max;
    double base_error;     /* i.i.d. sequencing error */
    double mate_frac;      /* fraction of fragments with a second mate entry */
    uint64_t seed;
} secedo_synth_spec;

int secedo_synth_generate(const secedo_synth_spec *spec, uint64_t *n_loci, uint64_t *n_entries,
                          uint32_t *chr_locus_off, uint32_t *locus_pos, uint64_t *locus_entry_off,
                          uint32_t *read_ids, uint32_t *id_base32);

#ifdef __cplusplus
}
#endif
#endif /* SECEDO_SYNTH_H */
