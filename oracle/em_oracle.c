/* em_oracle.c -- TEST INFRASTRUCTURE ONLY. See em_oracle.h. */
#include "em_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cluster_center (expectation_maximization.cpp:19-39) */
static int cluster_center(const uint32_t *id_base, uint64_t b, uint64_t e, const double *prob,
                          uint32_t n_cells, double theta, double center[4]) {
    center[0] = center[1] = center[2] = center[3] = 0.0;
    for (uint64_t i = b; i < e; ++i) { /* :23-25 */
        const uint32_t group = id_base[i] >> 2;
        if (group >= n_cells) return -1;
        center[id_base[i] & 3u] += prob[group];
    }
    double s = center[0] + center[1] + center[2] + center[3]; /* :27 */
    if (s == 0) {                                             /* :28-30 */
        for (int k = 0; k < 4; ++k) center[k] = log(0.25);
        return 0;
    }
    for (int k = 0; k < 4; ++k) center[k] = center[k] / s > theta ? center[k] / s : theta; /* :31-32 */
    s = center[0] + center[1] + center[2] + center[3];                                      /* :34 */
    for (int k = 0; k < 4; ++k) center[k] = log(center[k] / s);                             /* :35-36 */
    return 0;
}

int oracle_em(const uint32_t *chr_locus_off, uint32_t n_chr, const uint64_t *locus_entry_off,
              const uint32_t *id_base, const uint32_t *id_to_pos, uint32_t n_groups, double theta,
              double *prob_b, uint32_t n_cells, uint32_t max_iterations) {
    double *ll_a = calloc(n_cells ? n_cells : 1, sizeof(double));   /* :130-131, never reset */
    double *ll_b = calloc(n_cells ? n_cells : 1, sizeof(double));
    double *chr_a = malloc((n_cells ? n_cells : 1) * sizeof(double));
    double *chr_b = malloc((n_cells ? n_cells : 1) * sizeof(double));
    double *prob_a = malloc((n_cells ? n_cells : 1) * sizeof(double));
    int rc = 0;
    uint32_t iterations = 0;
    for (;;) {
        if (iterations == max_iterations) {
            rc = -3;
            break;
        }
        ++iterations;
        for (uint32_t c = 0; c < n_chr && rc == 0; ++c) { /* :135-147, one maximization_step per chromosome */
            for (uint32_t i = 0; i < n_cells; ++i) prob_a[i] = 1 - prob_b[i]; /* :62-65 */
            memset(chr_a, 0, n_cells * sizeof(double));                      /* :67-68 */
            memset(chr_b, 0, n_cells * sizeof(double));
            for (uint32_t l = chr_locus_off[c]; l < chr_locus_off[c + 1] && rc == 0; ++l) { /* :71-81 */
                const uint64_t b = locus_entry_off[l], e = locus_entry_off[l + 1];
                double center_a[4], center_b[4];
                if (cluster_center(id_base, b, e, prob_a, n_cells, theta, center_a)
                    || cluster_center(id_base, b, e, prob_b, n_cells, theta, center_b)) {
                    rc = -1;
                    break;
                }
                for (uint64_t i = b; i < e; ++i) { /* :77-80 */
                    const uint32_t group = id_base[i] >> 2;
                    if (group >= n_groups || id_to_pos[group] >= n_cells) {
                        rc = -2;
                        break;
                    }
                    chr_a[id_to_pos[group]] += center_a[id_base[i] & 3u];
                    chr_b[id_to_pos[group]] += center_b[id_base[i] & 3u];
                }
            }
            for (uint32_t i = 0; i < n_cells; ++i) { /* :142-145 */
                ll_a[i] += chr_a[i];
                ll_b[i] += chr_b[i];
            }
        }
        if (rc) break;
        /* expectation_step (:100-123) */
        double total = 0.0;
        for (uint32_t i = 0; i < n_cells; ++i) total += prob_b[i]; /* util.hpp:150-152 */
        const double prior_b = total / n_cells, prior_a = 1 - prior_b;
        int done = 1;
        for (uint32_t i = 0; i < n_cells; ++i) {
            double d = ll_b[i] - ll_a[i];
            d = d < -100. ? -100. : (d > 100. ? 100. : d);
            const double odds = exp(d);
            const double prob = 1 - 1 / (1 + odds * prior_b / prior_a);
            done &= fabs(prob - prob_b[i]) < 1e-2;
            prob_b[i] = prob;
        }
        if (done) break;
    }
    free(ll_a);
    free(ll_b);
    free(chr_a);
    free(chr_b);
    free(prob_a);
    return rc ? rc : (int)iterations;
}
