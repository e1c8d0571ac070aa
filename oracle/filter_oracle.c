/*
 * filter_oracle.c -- TEST INFRASTRUCTURE ONLY (see filter_oracle.h).
 * Plain-C restatement of the reference's locus filter; citations relative to /root/reference.
 */
#include "filter_oracle.h"

#include <fenv.h>
#include <math.h>
#include <stdlib.h>

/* Thresholds K for coverage 10, 20, ..., 200 (columns) and a 10-90 ... 50-50 split of the cells
 * (rows): numeric constants of the algorithm, util/is_significant.cpp:11-47 (generated upstream by
 * scripts/K.r; they cannot be re-derived here, R is not available). */
static const double K_TABLE[5][20] = {
    { -1.64504967001201, -1.38868450353301, -1.38780664765677, -1.38779600211955, -1.3877952855556,
      -1.38779524274215, -1.38779524274142, -1.38779524274141, -1.3877952427414, -1.38779524274139,
      -1.38779524274138, -1.38779524274138, -1.38779524274138, -1.38779524274138, -1.38779524274139,
      -1.38780870444455, -1.38780870444455, -1.38780870444455, -1.38780870444455, -1.38780870444455 },
    { -1.56013904495168, -1.38819451352203, -1.38781438946096, -1.38779659244035, -1.38779537799054,
      -1.3877952484612, -1.3877952427842, -1.38779524274906, -1.3877952427457, -1.38779524274275,
      -1.38780870444458, -1.38780870444459, -1.38780870444459, -1.38780870444469, -1.38780870444459,
      -1.42736056742577, -1.42736056742575, -6.19144172018466, -6.19144172018466, -14.1885779508362 },
    { -1.47780038365618, -1.3885722463397, -1.38781428162649, -1.3877984410546, -1.38779548312685,
      -1.3877952855556, -1.38779524455204, -1.38779524331456, -1.38780873675669, -1.38780870687333,
      -1.42737804009806, -6.19144172131432, -14.1885779508648, -6.1914418045659, -30.1993093269287,
      -30.1993093268559, -30.1993093268539, -54.2154105288032, -62.2207775961199, -46.2100434614866 },
    { -1.47780038365618, -1.38868450353301, -1.38782829051844, -1.3877984410546, -1.38779625512927,
      -1.38779556321717, -1.38780972304588, -1.3878087226245, -6.21747860711653, -22.1939432034943,
      -14.1886670526002, -22.1939422903721, -46.2100434614866, -54.2154105288069, -70.2261446634366,
      -62.2207775961199, -86.2368787980699, -110.25298000002, -118.258347067337, -102.247612932703 },
    { -1.52859626647315, -1.38967447346712, -1.38787138908447, -1.38780282263764, -1.387805349423,
      -1.38882047800373, -1.49793700616569, -6.19975747800726, -22.197881249831, -38.2046765807324,
      -38.2046769835162, -70.2261446634383, -54.2154105303641, -78.2315117307532, -86.2368787980699,
      -118.258347067337, -126.263714134653, -134.26908120197, -158.28518240392, -158.28518240392 },
};

static int cmp_u16(const void *a, const void *b) {
    return (int)*(const uint16_t *)a - (int)*(const uint16_t *)b;
}

/* returns 1 when the cheap integer tests already decide "not significant" */
static int rejected_early(uint16_t c[4], uint32_t *coverage) {
    *coverage = (uint32_t)c[0] + c[1] + c[2] + c[3];
    if (*coverage < 2) return 1;                 /* :83-85 */
    qsort(c, 4, sizeof(uint16_t), cmp_u16);      /* :88 ascending */
    if (c[2] == 0) return 1;                     /* :90-92 all bases equal */
    if (c[2] + c[1] + c[0] < 5) return 1;        /* :97-99 */
    if (c[3] < 1.5 * c[2]) return 1;             /* :101-103 */
    return 0;
}

void oracle_significance_terms(const uint16_t base_count[4], double theta, uint32_t cell_proportion,
                               double *statistic, double *threshold) {
    uint16_t c[4] = { base_count[0], base_count[1], base_count[2], base_count[3] };
    uint32_t coverage;
    if (rejected_early(c, &coverage)) {
        *statistic = NAN;
        *threshold = NAN;
        return;
    }
    /* priors, :52-57 (log_homo_prior is log(hetero_prior) in the reference: kept as is) */
    const double hetero_prior = 0.0005, mut_prior = 1e-6;
    const double homo_prior = 1 - hetero_prior - mut_prior;
    const double log_homo_prior = log(hetero_prior);
    const double log_1_4 = log(1. / 4);
    /* :106-107 threshold column: coverage / 10 rounded to nearest even, minus 1, clamped to 0..19 */
    fesetround(FE_TONEAREST);
    double col = nearbyint(coverage / 10.) - 1;
    if (col < 0.) col = 0.;
    if (col > 19.) col = 19.;
    const uint32_t idx = (uint32_t)col;
    /* :110-116 */
    double log_prob_homozygous = c[3] * log(1 - theta) + (coverage - c[3]) * log(theta / 3);
    log_prob_homozygous += log_1_4;
    log_prob_homozygous += log_homo_prior;
    /* :120-135 the five hypotheses of the evidence */
    const double prob_all_c1 = homo_prior * pow(1 - theta, c[3]) * pow(theta / 3, coverage - c[3]);
    const double prob_hetero = hetero_prior * pow(0.5 - theta / 3, c[3] + c[2]) * pow(theta / 3, c[0] + c[1]);
    const double prob_homo_som = homo_prior * mut_prior * pow(0.75 - 2 * theta / 3, c[3]) * pow(0.25, c[2])
            * pow(theta / 3, c[0] + c[1]);
    const double prob_hetero_som = hetero_prior * mut_prior * pow(0.5 - theta, c[3]) * pow(0.25, c[1] + c[2])
            * pow(theta / 3, c[0]);
    const double prob_two_somatic = hetero_prior * mut_prior * mut_prior * pow(1 - theta, coverage);
    const double log_evidence
            = log(prob_all_c1 + prob_hetero + prob_homo_som + prob_hetero_som + prob_two_somatic);
    *statistic = log_prob_homozygous - log_evidence;
    *threshold = K_TABLE[cell_proportion][idx];
}

int oracle_is_significant(const uint16_t base_count[4], double theta, uint32_t cell_proportion) {
    double s, k;
    oracle_significance_terms(base_count, theta, cell_proportion, &s, &k);
    if (isnan(k)) return 0;
    return s < k; /* :137 */
}

int oracle_filter(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
                  const uint64_t *locus_entry_off, const uint32_t *read_ids, const uint32_t *id_base,
                  const uint32_t *id_to_pos, uint32_t n_groups, double theta, uint32_t cell_proportion,
                  uint32_t *out_chr_locus_off, uint32_t *out_locus_pos, uint64_t *out_locus_entry_off,
                  uint32_t *out_read_ids, uint32_t *out_id_base, uint64_t *out_n_loci,
                  uint64_t *out_n_entries, double *avg_coverage) {
    uint64_t nl = 0, ne = 0;
    (void)n_groups;
    out_chr_locus_off[0] = 0;
    out_locus_entry_off[0] = 0;
    for (uint32_t c = 0; c < n_chr; ++c) {
        for (uint32_t l = chr_locus_off[c]; l < chr_locus_off[c + 1]; ++l) { /* :162 */
            uint16_t count[4] = { 0, 0, 0, 0 };
            const uint64_t first = ne;
            for (uint64_t e = locus_entry_off[l]; e < locus_entry_off[l + 1]; ++e) { /* :167-174 */
                if (id_to_pos[id_base[e] >> 2] == ORACLE_NO_POS) continue;
                out_read_ids[ne] = read_ids[e];
                out_id_base[ne] = id_base[e];
                ++ne;
                count[id_base[e] & 3u]++;
            }
            if (oracle_is_significant(count, theta, cell_proportion)) { /* :176-180 */
                out_locus_pos[nl] = locus_pos[l];
                ++nl;
                out_locus_entry_off[nl] = ne;
            } else {
                ne = first; /* drop the locus */
            }
        }
        out_chr_locus_off[c + 1] = (uint32_t)nl;
    }
    *out_n_loci = nl;
    *out_n_entries = ne;
    /* :186-188: total coverage is accumulated in uint32 in the reference */
    *avg_coverage = nl == 0 ? 0 : (double)(uint32_t)ne / (double)(uint32_t)nl;
    return 0;
}
