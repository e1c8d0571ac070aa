/*
 * simmat_oracle.c -- TEST INFRASTRUCTURE ONLY (see simmat_oracle.h).
 *
 * Plain-C restatement of the reference algorithm for the similarity-matrix path.
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference). Serial: the reference's OpenMP region only changes the order in
 * which equal-valued terms are summed (1e-13 relative), while
 * num_threads stays a semantic input through the flush threshold.
 *
 * Parity pin: oracle/_ref (the compiled reference) + tests/golden (vectors generated
 * from it). See tests/test_oracle_vs_ref.py and tests/test_oracle_golden.py.
 */
#include "simmat_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* Probability tables: similarity_matrix.cpp:38-104 (struct Cache)                      */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    uint32_t size; /* number of entries per power table == number of Pascal rows */
    double *p_ss, *p_sd, *p_ds, *p_dd;         /* pow_p_same_same ... pow_p_diff_diff */
    double *one_h_eps, *one_h_eps2, *h_eps2;   /* (1-eps-h)^k, (1-eps/2-h)^k, (h+eps/2)^k */
    double *hh, *eps, *half;                   /* h^k, eps^k, 0.5^k */
    double *pss_pds, *psd_pdd;                 /* (p_ss+p_ds)^k, (p_sd+p_dd)^k */
    uint64_t **comb;                           /* Pascal triangle, u64 wrap-around kept */
    long double **combl;                       /* the same triangle without the wrap (see below) */
} Tables;

/* The reference's u64 binomials and their u64 products wrap once x_s + x_d reaches about 48
 * (e.g. D(50,3) is off by 3e-6, D(60,4) by 0.11): its output there is an artefact of the wrap, not
 * the value of its own formula. oracle_set_exact_binomials(1) evaluates the SAME nested sums with a
 * long-double Pascal triangle, i.e. the reference formula in exact arithmetic; the default (0)
 * reproduces the reference bit for bit. */
static _Thread_local int g_exact_mode = 0;
void oracle_set_exact_binomials(int on) { g_exact_mode = on; }
/* Mode 2: exact binomials only for x_s + x_d > 128 -- what the MI355X path documents for read pairs that
 * share more loci than its table of reference-identical terms covers. */
#define g_exact_binomials (g_exact_mode == 1 || (g_exact_mode == 2 && x_s + x_d > 128))

static double *pow_table(double base, uint32_t size) {
    /* similarity_matrix.cpp:53-65 start each table as {1, x}; :85-94 extend by
     * a.back() * a[1] for p = 2 .. max_read_size-1. */
    uint32_t n = size < 2 ? 2 : size;
    double *t = (double *)malloc(sizeof(double) * n);
    t[0] = 1;
    t[1] = base;
    for (uint32_t p = 2; p < n; ++p) {
        t[p] = t[p - 1] * t[1];
    }
    return t;
}

static void tables_init(Tables *t, double epsilon, double h, double theta, uint32_t size) {
    /* similarity_matrix.cpp:43-51 */
    const double theta2 = theta * theta;
    const double p_same_diff = 2 * theta * (1 - theta) + 2 * theta2 / 3;
    const double p_same_same = 1 - p_same_diff;
    const double p_diff_same = 2 * (1 - theta) * theta / 3 + 2 * theta2 / 9;
    const double p_diff_diff = 1 - p_diff_same;

    t->size = size < 2 ? 2 : size;
    t->p_ss = pow_table(p_same_same, size);
    t->p_sd = pow_table(p_same_diff, size);
    t->p_ds = pow_table(p_diff_same, size);
    t->p_dd = pow_table(p_diff_diff, size);
    t->one_h_eps = pow_table(1 - epsilon - h, size);
    t->one_h_eps2 = pow_table(1 - epsilon * 0.5 - h, size);
    t->h_eps2 = pow_table(h + epsilon * 0.5, size);
    t->hh = pow_table(h, size);
    t->eps = pow_table(epsilon, size);
    t->half = pow_table(0.5, size);
    t->pss_pds = pow_table(p_same_same + p_diff_same, size);
    t->psd_pdd = pow_table(p_same_diff + p_diff_diff, size);

    /* similarity_matrix.cpp:67 and :95-101: rows 0 and 1 given, row p has p+1 entries,
     * unsigned 64-bit additions (they wrap silently from row 68 on). */
    t->comb = (uint64_t **)malloc(sizeof(uint64_t *) * t->size);
    t->comb[0] = (uint64_t *)malloc(sizeof(uint64_t));
    t->comb[0][0] = 1;
    t->comb[1] = (uint64_t *)malloc(2 * sizeof(uint64_t));
    t->comb[1][0] = 1;
    t->comb[1][1] = 1;
    for (uint32_t p = 2; p < t->size; ++p) {
        uint64_t *row = (uint64_t *)malloc(sizeof(uint64_t) * (p + 1));
        const uint64_t *prev = t->comb[p - 1];
        row[0] = 1;
        row[p] = 1;
        for (uint32_t i = 1; i < p; ++i) {
            row[i] = prev[i - 1] + prev[i];
        }
        t->comb[p] = row;
    }
    t->combl = (long double **)malloc(sizeof(long double *) * t->size);
    for (uint32_t p = 0; p < t->size; ++p) {
        t->combl[p] = (long double *)malloc(sizeof(long double) * (p + 1));
        t->combl[p][0] = 1;
        t->combl[p][p] = 1;
        for (uint32_t i = 1; i < p; ++i) t->combl[p][i] = t->combl[p - 1][i - 1] + t->combl[p - 1][i];
    }
}

static void tables_free(Tables *t) {
    free(t->p_ss); free(t->p_sd); free(t->p_ds); free(t->p_dd);
    free(t->one_h_eps); free(t->one_h_eps2); free(t->h_eps2);
    free(t->hh); free(t->eps); free(t->half); free(t->pss_pds); free(t->psd_pdd);
    for (uint32_t p = 0; p < t->size; ++p) { free(t->comb[p]); free(t->combl[p]); }
    free(t->comb);
    free(t->combl);
}

/* similarity_matrix.cpp:153-170 (without the memo: the callers memoise). The factor
 * order of the reference expression is kept so that the rounding is the same; the two
 * binomials are multiplied as u64 before the product becomes a double. */
static double eval_log_prob_same(uint32_t x_s, uint32_t x_d, const Tables *c) {
    double p = 0;
    for (uint32_t k = 0; k <= x_s; ++k) {
        for (uint32_t l = 0; l <= x_d; ++l) {
            const double binom = g_exact_binomials
                    ? (double)(c->combl[x_s][k] * c->combl[x_d][l])
                    : (double)(c->comb[x_s][k] * c->comb[x_d][l]);
            p += binom * c->one_h_eps2[k + l] * 0.5
                    * (c->p_ss[k] * c->p_sd[l] + c->p_ds[k] * c->p_dd[l])
                    * c->h_eps2[x_s + x_d - k - l] * c->p_ss[x_s - k] * c->p_sd[x_d - l];
        }
    }
    p *= g_exact_binomials ? (double)c->combl[x_s + x_d][x_s] : (double)c->comb[x_s + x_d][x_s];
    return log(p);
}

/* similarity_matrix.cpp:117-141 (without the memo). Four nested sums; the four binomials
 * are multiplied as u64 first, exactly as the reference's left-to-right product does. */
static double eval_log_prob_diff(uint32_t x_s, uint32_t x_d, const Tables *c) {
    double prob = 0;
    for (uint32_t k = 0; k <= x_s; ++k) {
        for (uint32_t l = 0; l <= x_d; ++l) {
            for (uint32_t p = 0; p <= x_s - k; ++p) {
                for (uint32_t q = 0; q <= x_d - l; ++q) {
                    const double binom = g_exact_binomials
                            ? (double)(c->combl[x_s][k] * c->combl[x_d][l] * c->combl[x_s - k][p]
                                       * c->combl[x_d - l][q])
                            : (double)(c->comb[x_s][k] * c->comb[x_d][l] * c->comb[x_s - k][p]
                                       * c->comb[x_d - l][q]);
                    uint32_t rest = x_s + x_d - k - l - p - q;
                    prob += binom * c->one_h_eps[k + l] * 0.5
                            * (c->p_ss[k] * c->p_sd[l] + c->p_ds[k] * c->p_dd[l])
                            * c->eps[rest] * c->half[rest] * c->pss_pds[x_s - k - p]
                            * c->psd_pdd[x_d - l - q] * c->hh[p + q] * c->p_ss[p] * c->p_sd[q];
                }
            }
        }
    }
    prob *= g_exact_binomials ? (double)c->combl[x_s + x_d][x_s] : (double)c->comb[x_s + x_d][x_s];
    return log(prob);
}

double oracle_log_prob_same(uint32_t x_s, uint32_t x_d, double eps, double h, double theta,
                            uint32_t table_size) {
    Tables t;
    tables_init(&t, eps, h, theta, table_size);
    double v = (x_s + x_d < t.size) ? eval_log_prob_same(x_s, x_d, &t) : NAN;
    tables_free(&t);
    return v;
}

double oracle_log_prob_diff(uint32_t x_s, uint32_t x_d, double eps, double h, double theta,
                            uint32_t table_size) {
    Tables t;
    tables_init(&t, eps, h, theta, table_size);
    double v = (x_s + x_d < t.size) ? eval_log_prob_diff(x_s, x_d, &t) : NAN;
    tables_free(&t);
    return v;
}

/* ------------------------------------------------------------------------------------ */
/* Normalisation: similarity_matrix.cpp:271-293 with the Mat<T> operations it calls      */
/* (util/mat.hpp:184-206 scalar ops, :388-413 exp/inv/min/max, :479-483 fill_diagonal)   */
/* ------------------------------------------------------------------------------------ */

int oracle_normalize(int normalization, double *m, uint32_t n) {
    const uint64_t nn = (uint64_t)n * n;
    switch (normalization) {
        case ORACLE_ADD_MIN: {
            /* sim_mat *= -1; sim_mat += |min(sim_mat)|  (min over all n*n entries) */
            for (uint64_t i = 0; i < nn; ++i) m[i] *= -1;
            double mn = nn ? m[0] : 0;
            for (uint64_t i = 1; i < nn; ++i) {
                if (m[i] < mn) mn = m[i];
            }
            const double add = fabs(mn);
            for (uint64_t i = 0; i < nn; ++i) m[i] += add;
            break;
        }
        case ORACLE_EXPONENTIATE:
            /* 1 / (1 + exp(m)), three element-wise passes in the reference */
            for (uint64_t i = 0; i < nn; ++i) m[i] = exp(m[i]);
            for (uint64_t i = 0; i < nn; ++i) m[i] += 1;
            for (uint64_t i = 0; i < nn; ++i) m[i] = 1. / m[i];
            break;
        case ORACLE_SCALE_MAX_1: {
            /* diagonal to zero first, then multiply by 1/max (no exp, despite the comment
             * at similarity_matrix.cpp:287) */
            for (uint32_t i = 0; i < n; ++i) m[(uint64_t)i * n + i] = 0;
            double mx = nn ? m[0] : 0;
            for (uint64_t i = 1; i < nn; ++i) {
                if (m[i] > mx) mx = m[i];
            }
            const double f = 1. / mx;
            for (uint64_t i = 0; i < nn; ++i) m[i] *= f;
            break;
        }
        default: return -2; /* the reference throws std::logic_error (similarity_matrix.cpp:264) */
    }
    for (uint32_t i = 0; i < n; ++i) m[(uint64_t)i * n + i] = 0; /* :292 */
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Live reads: similarity_matrix.cpp:172-182 (struct Read) + the deque / hash map of      */
/* :325-343. Reads are kept in first-appearance order in one array; `front` plays the     */
/* role of the deque's begin after flushed reads were popped (:368-371).                  */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    uint32_t *pos;
    uint8_t *base;
    uint32_t n, cap;
    uint32_t group;
    uint32_t start;
    uint32_t read_id;
} Rd;

typedef struct {
    Rd *r;
    uint32_t n, cap, front;
    /* read_id -> index in r; an entry whose index is < front is a flushed (erased) read */
    uint32_t *slot_key, *slot_val;
    uint8_t *slot_used;
    uint32_t n_slots;
} Live;

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

static void live_reset(Live *lv, uint64_t expected_entries) {
    for (uint32_t i = 0; i < lv->n; ++i) { free(lv->r[i].pos); free(lv->r[i].base); }
    lv->n = 0;
    lv->front = 0;
    uint64_t want = 16;
    while (want < expected_entries * 2 + 2) want <<= 1;
    if (want > lv->n_slots) {
        free(lv->slot_key); free(lv->slot_val); free(lv->slot_used);
        lv->n_slots = (uint32_t)want;
        lv->slot_key = (uint32_t *)malloc(sizeof(uint32_t) * want);
        lv->slot_val = (uint32_t *)malloc(sizeof(uint32_t) * want);
        lv->slot_used = (uint8_t *)malloc(want);
    }
    memset(lv->slot_used, 0, lv->n_slots);
}

/* returns the slot of read_id (used or the free slot where it belongs) */
static uint32_t live_slot(const Live *lv, uint32_t read_id) {
    uint32_t mask = lv->n_slots - 1;
    uint32_t s = mix32(read_id) & mask;
    while (lv->slot_used[s] && lv->slot_key[s] != read_id) s = (s + 1) & mask;
    return s;
}

static Rd *live_push(Live *lv) {
    if (lv->n == lv->cap) {
        lv->cap = lv->cap ? lv->cap * 2 : 1024;
        lv->r = (Rd *)realloc(lv->r, sizeof(Rd) * lv->cap);
    }
    Rd *rd = &lv->r[lv->n++];
    memset(rd, 0, sizeof(*rd));
    return rd;
}

static void rd_append(Rd *rd, uint32_t position, uint8_t base) {
    if (rd->n == rd->cap) {
        rd->cap = rd->cap ? rd->cap * 2 : 4;
        rd->pos = (uint32_t *)realloc(rd->pos, sizeof(uint32_t) * rd->cap);
        rd->base = (uint8_t *)realloc(rd->base, rd->cap);
    }
    rd->pos[rd->n] = position;
    rd->base[rd->n] = base;
    rd->n++;
}

/* ------------------------------------------------------------------------------------ */

typedef struct {
    double *mat_same, *mat_diff;   /* similarity_matrix.cpp:306-307 */
    double *lp_same, *lp_diff;     /* memo tables, :314-317 */
    uint32_t lp_dim;
    const Tables *tables;
    const uint32_t *g2p;
    uint32_t n_cells;
    uint64_t updates, pairs;
    int out_of_tables; /* a read pair shared >= max_fragment_length loci (see compare_with_later_reads) */
} Acc;

static _Thread_local uint64_t g_last_updates, g_last_pairs;

/* The reference sums log P_same and log P_diff in two matrices and subtracts at the end (:428). With
 * very many pairs per cell pair (hundreds of thousands) the two sums reach ~1e6 while their
 * difference stays ~1e2, and the subtraction loses digits: the reference's
 * own output is then only good to ~1e-8 relative. oracle_set_direct_llr_sum(1) adds
 * (log P_diff - log P_same) per pair into mat_diff instead -- the same terms without the
 * cancellation -- so that a test can tell the reference's rounding noise from a real discrepancy.
 * Default 0 = the reference's arithmetic. Thread-local. */
static _Thread_local int g_direct_llr_sum = 0;
void oracle_set_direct_llr_sum(int on) { g_direct_llr_sum = on; }
uint64_t oracle_last_updates(void) { return g_last_updates; }
uint64_t oracle_last_read_pairs(void) { return g_last_pairs; }

static double memo_same(Acc *a, uint32_t x_s, uint32_t x_d) {
    double *slot = &a->lp_same[(uint64_t)x_s * a->lp_dim + x_d];
    if (*slot == DBL_MAX) *slot = eval_log_prob_same(x_s, x_d, a->tables);
    return *slot;
}

static double memo_diff(Acc *a, uint32_t x_s, uint32_t x_d) {
    double *slot = &a->lp_diff[(uint64_t)x_s * a->lp_dim + x_d];
    if (*slot == DBL_MAX) *slot = eval_log_prob_diff(x_s, x_d, a->tables);
    return *slot;
}

/* similarity_matrix.cpp:189-243 (compare_with_reads) fused with :246-254 (apply_updates):
 * the reference queues (i, j, v) per thread and applies them after the parallel loop; the
 * serial restatement applies them at once. */
static void compare_with_later_reads(Acc *a, const Live *lv, uint32_t first) {
    const Rd *r1 = &lv->r[first];
    if (r1->n == 0) return; /* :200 */
    const uint32_t index1 = a->g2p[r1->group];
    for (uint32_t j = first + 1; j < lv->n; ++j) {
        const Rd *r2 = &lv->r[j];
        if (r2->n == 0) continue; /* :205 */
        const uint32_t index2 = a->g2p[r2->group];
        if (index1 == index2 || r1->pos[r1->n - 1] < r2->pos[0]) continue; /* :215 */
        uint32_t x_s = 0, x_d = 0;
        for (uint32_t i1 = 0, i2 = 0; i1 < r1->n && i2 < r2->n;) { /* :223-229 */
            if (r1->pos[i1] == r2->pos[i2]) {
                if (r1->base[i1] == r2->base[i2]) x_s++; else x_d++;
                i1++; i2++;
            } else if (r1->pos[i1] < r2->pos[i2]) {
                i1++;
            } else {
                i2++;
            }
        }
        if (x_s == 0 && x_d == 0) continue; /* :231 */
        /* The reference's tables have max_fragment_length rows (:314-317, :330): a read pair that shares
         * that many loci makes it read past them (undefined behaviour there). No answer exists to restate:
         * the call fails with ORACLE_E_OUT_OF_TABLES instead of indexing out of bounds. */
        if (x_s + x_d >= a->tables->size) {
            a->out_of_tables = 1;
            continue;
        }
        a->updates += x_s + x_d;
        a->pairs += 1;
        const uint64_t ij = (uint64_t)index1 * a->n_cells + index2;
        const uint64_t ji = (uint64_t)index2 * a->n_cells + index1;
        if (g_direct_llr_sum) {
            a->mat_diff[ij] += memo_diff(a, x_s, x_d) - memo_same(a, x_s, x_d);
            a->mat_diff[ji] = a->mat_diff[ij];
            continue;
        }
        a->mat_same[ij] += memo_same(a, x_s, x_d); /* :249 */
        a->mat_same[ji] = a->mat_same[ij];         /* :250 */
        a->mat_diff[ij] += memo_diff(a, x_s, x_d);
        a->mat_diff[ji] = a->mat_diff[ij];
    }
}

int oracle_simmat_compute(const uint32_t *chr_locus_off, uint32_t n_chr,
                          const uint32_t *locus_pos, const uint64_t *locus_entry_off,
                          const uint32_t *read_ids, const uint32_t *id_base,
                          const uint32_t *g2p, uint32_t n_groups, uint32_t num_cells,
                          uint32_t mfl, double mutation_rate, double homozygous_rate,
                          double seq_error_rate, uint32_t num_threads, int normalization,
                          double *out, double *out_raw) {
    if (normalization < 0 || normalization > 2) return -2;
    if (num_threads == 0 || mfl < 2) return -1;
    const uint64_t nn = (uint64_t)num_cells * num_cells;

    Tables tables;
    tables_init(&tables, mutation_rate, homozygous_rate, seq_error_rate, mfl); /* :330 */

    Acc a;
    memset(&a, 0, sizeof(a));
    a.mat_same = (double *)calloc(nn ? nn : 1, sizeof(double));
    a.mat_diff = (double *)calloc(nn ? nn : 1, sizeof(double));
    a.lp_dim = tables.size;
    a.lp_same = (double *)malloc(sizeof(double) * a.lp_dim * a.lp_dim);
    a.lp_diff = (double *)malloc(sizeof(double) * a.lp_dim * a.lp_dim);
    for (uint64_t i = 0; i < (uint64_t)a.lp_dim * a.lp_dim; ++i) {
        a.lp_same[i] = DBL_MAX;
        a.lp_diff[i] = DBL_MAX;
    }
    a.tables = &tables;
    a.g2p = g2p;
    a.n_cells = num_cells;
    (void)n_groups;

    Live lv;
    memset(&lv, 0, sizeof(lv));

    /* `completed` counts, from the front of the live list, the reads that started at least
     * max_fragment_length before the current locus; it is NOT reset between chromosomes
     * (:344 is outside the chromosome loop). */
    uint32_t completed = 0;
    const uint32_t batch = 4; /* :354 */
    for (uint32_t c = 0; c < n_chr; ++c) {
        const uint32_t l0 = chr_locus_off[c], l1 = chr_locus_off[c + 1];
        live_reset(&lv, l1 > l0 ? locus_entry_off[l1] - locus_entry_off[l0] : 0);
        for (uint32_t l = l0; l < l1; ++l) {
            const uint32_t position = locus_pos[l];
            /* :348-352 (uint32 arithmetic, as in the reference) */
            for (uint32_t i = completed; i < lv.n - lv.front
                 && (uint32_t)(lv.r[lv.front + i].start + mfl) <= position; ++i) {
                ++completed;
            }
            /* :356-373 flush: compare every completed read with all later live reads, then
             * drop the completed ones from the live list */
            if (completed >= batch * num_threads) {
                for (uint32_t i = 0; i < completed; ++i) {
                    compare_with_later_reads(&a, &lv, lv.front + i);
                }
                lv.front += completed; /* erased: a later entry with the same id starts anew */
                completed = 0;
            }
            /* :376-403 add this locus' bases to the live reads */
            for (uint64_t e = locus_entry_off[l]; e < locus_entry_off[l + 1]; ++e) {
                const uint32_t rid = read_ids[e];
                const uint8_t base = (uint8_t)(id_base[e] & 3u);
                const uint32_t group = id_base[e] >> 2;
                const uint32_t s = live_slot(&lv, rid);
                const int live = lv.slot_used[s] && lv.slot_val[s] >= lv.front;
                if (!live) { /* :379-382 a new read starts */
                    Rd *rd = live_push(&lv);
                    rd->group = group;
                    rd->start = position;
                    rd->read_id = rid;
                    rd_append(rd, position, base);
                    lv.slot_used[s] = 1;
                    lv.slot_key[s] = rid;
                    lv.slot_val[s] = lv.n - 1;
                } else {
                    Rd *rd = &lv.r[lv.slot_val[s]];
                    if (rd->n && rd->pos[rd->n - 1] == position) { /* :387-395 */
                        if (rd->base[rd->n - 1] != base) rd->n--; /* conflicting mates */
                        continue;
                    }
                    rd_append(rd, position, base); /* :400-401 */
                }
            }
        }
        /* :407-408: the live list is cleared at the end of a chromosome, so the loop at
         * :412-418 never sees a read -- reads not flushed by now never act as the earlier
         * read of a pair. */
    }
    live_reset(&lv, 0);
    free(lv.r); free(lv.slot_key); free(lv.slot_val); free(lv.slot_used);

    /* :428 */
    for (uint64_t i = 0; i < nn; ++i) a.mat_diff[i] -= a.mat_same[i];
    if (out_raw) memcpy(out_raw, a.mat_diff, sizeof(double) * nn);
    int rc = oracle_normalize(normalization, a.mat_diff, num_cells); /* :430 */
    memcpy(out, a.mat_diff, sizeof(double) * nn);

    g_last_updates = a.updates;
    g_last_pairs = a.pairs;
    free(a.mat_same); free(a.mat_diff); free(a.lp_same); free(a.lp_diff);
    tables_free(&tables);
    return a.out_of_tables ? -3 : rc;
}
