#include "llr_table.hpp"

#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <thread>

namespace secedo {

LlrModel make_llr_model(double eps, double h, double theta) {
    // read-pair probabilities (reference: similarity_matrix.cpp:43-51), in long double so that the
    // logs below are correctly rounded doubles
    const long double t = theta, e = eps, hh = h;
    const long double psd = 2 * t * (1 - t) + 2 * t * t / 3;          // same letters read as different
    const long double pss = 1 - psd;
    const long double pds = 2 * (1 - t) * t / 3 + 2 * t * t / 9;      // different letters read as same
    const long double pdd = 1 - pds;

    // different genotypes (:117-141): per matching locus the three summands are
    // (1-eps-h)*{pss|pds} + h*pss + (eps/2)*(pss+pds); per mismatching locus the same with
    // {psd|pdd}, psd, (psd+pdd)
    const long double a = 1 - e - hh;
    const long double u1 = a * pss + hh * pss + e / 2 * (pss + pds);
    const long double v1 = a * psd + hh * psd + e / 2 * (psd + pdd);
    const long double u2 = a * pds + hh * pss + e / 2 * (pss + pds);
    const long double v2 = a * pdd + hh * psd + e / 2 * (psd + pdd);
    // same genotype (:153-170): (1-eps/2-h)*{pss|pds} + (h+eps/2)*pss, and the psd/pdd analogue
    const long double a2 = 1 - e / 2 - hh, b = hh + e / 2;
    const long double w1 = a2 * pss + b * pss;
    const long double z1 = a2 * psd + b * psd;
    const long double w2 = a2 * pds + b * pss;
    const long double z2 = a2 * pdd + b * psd;

    LlrModel m;
    m.ln_u1 = static_cast<double>(std::log(u1));
    m.ln_v1 = static_cast<double>(std::log(v1));
    m.ln_u2 = static_cast<double>(std::log(u2));
    m.ln_v2 = static_cast<double>(std::log(v2));
    m.ln_w1 = static_cast<double>(std::log(w1));
    m.ln_z1 = static_cast<double>(std::log(z1));
    m.ln_w2 = static_cast<double>(std::log(w2));
    m.ln_z2 = static_cast<double>(std::log(z2));
    return m;
}

namespace {
// log(exp(a) + exp(b))
inline long double log_add(long double a, long double b) {
    const long double hi = std::max(a, b), lo = std::min(a, b);
    if (std::isinf(hi)) return hi;
    return hi + std::log1p(std::exp(lo - hi));
}
}  // namespace

double llr(const LlrModel &m, uint32_t x_s, uint32_t x_d) {
    const long double s = x_s, d = x_d;
    const long double diff = log_add(s * m.ln_u1 + d * m.ln_v1, s * m.ln_u2 + d * m.ln_v2);
    const long double same = log_add(s * m.ln_w1 + d * m.ln_z1, s * m.ln_w2 + d * m.ln_z2);
    return static_cast<double>(diff - same);
}

// ---- the reference's own evaluation --------------------------------------------------------------
namespace {

constexpr uint32_t kRows = kLlrRefMax + 1;

// Powers as the reference builds them (each entry = previous * base, similarity_matrix.cpp:85-94) and
// Pascal's triangle in uint64_t (:95-101). Up to row 67 no single binomial wraps, the PRODUCTS do from about
// row 48 on; beyond row 67 the additions of the triangle wrap too (unsigned arithmetic modulo 2^64 here as there).
// `rows` = 1 + the largest x_s + x_d wanted (the reference sizes its tables by max_fragment_length, :67-101; an
// entry does not depend on how far the tables go on behind it).
struct RefTables {
    uint32_t rows;
    std::vector<double> pss, psd, pds, pdd;
    std::vector<double> a1, a2, b2, hh, ehalf;  // (1-e-h)^k (1-e/2-h)^k (h+e/2)^k h^k (e^k * .5^k)
    std::vector<double> sum_s, sum_d;           // (pss+pds)^k, (psd+pdd)^k
    std::vector<uint64_t> comb_;                // rows x rows
    const uint64_t *comb(uint32_t n) const { return comb_.data() + (size_t)n * rows; }

    RefTables(double eps, double h, double theta, uint32_t n_rows = kRows) : rows(std::max(n_rows, 2u)) {
        const double t2 = theta * theta;
        const double p_sd = 2 * theta * (1 - theta) + 2 * t2 / 3;  // :45
        const double p_ss = 1 - p_sd;                              // :47
        const double p_ds = 2 * (1 - theta) * theta / 3 + 2 * t2 / 9;  // :49
        const double p_dd = 1 - p_ds;                              // :51
        auto powers = [&](std::vector<double> &out, double base) {
            out.assign(rows, 0.0);
            out[0] = 1;
            out[1] = base;
            for (uint32_t k = 2; k < rows; ++k) out[k] = out[k - 1] * base;
        };
        powers(pss, p_ss);
        powers(psd, p_sd);
        powers(pds, p_ds);
        powers(pdd, p_dd);
        powers(a1, 1 - eps - h);
        powers(a2, 1 - eps * 0.5 - h);
        powers(b2, h + eps * 0.5);
        powers(hh, h);
        powers(sum_s, p_ss + p_ds);
        powers(sum_d, p_sd + p_dd);
        std::vector<double> e, half;
        powers(e, eps);
        powers(half, 0.5);
        ehalf.assign(rows, 0.0);
        for (uint32_t k = 0; k < rows; ++k) ehalf[k] = e[k] * half[k];  // the reference multiplies the two (:131-132)
        comb_.assign((size_t)rows * rows, 0);
        for (uint32_t n = 0; n < rows; ++n) {
            uint64_t *row = comb_.data() + (size_t)n * rows;
            row[0] = row[n] = 1;
            for (uint32_t i = 1; i < n; ++i) row[i] = comb(n - 1)[i - 1] + comb(n - 1)[i];
        }
    }

    // :153-170. Terms are positive, so the order of the floating-point factors moves the sum by a few
    // ulp only; the integer product is what must be reproduced exactly.
    double log_same(uint32_t xs, uint32_t xd) const {
        double p = 0;
        const uint64_t *cs = comb(xs), *cd = comb(xd);
        for (uint32_t k = 0; k <= xs; ++k) {
            for (uint32_t l = 0; l <= xd; ++l) {
                const uint64_t c = cs[k] * cd[l];
                p += static_cast<double>(c) * a2[k + l] * 0.5 * (pss[k] * psd[l] + pds[k] * pdd[l])
                        * b2[xs + xd - k - l] * pss[xs - k] * psd[xd - l];
            }
        }
        p *= static_cast<double>(comb(xs + xd)[xs]);
        return std::log(p);
    }

    // the share of k = k_begin .. k_end - 1 of the four-fold sum of :117-141 (before the last factor and the log)
    double diff_part(uint32_t xs, uint32_t xd, uint32_t k_begin, uint32_t k_end) const {
        double prob = 0;
        const uint64_t *cs = comb(xs), *cd = comb(xd);
        for (uint32_t k = k_begin; k < k_end; ++k) {
            for (uint32_t l = 0; l <= xd; ++l) {
                const uint64_t ckl = cs[k] * cd[l];
                const double f = a1[k + l] * 0.5 * (pss[k] * psd[l] + pds[k] * pdd[l]);
                double inner = 0;
                const uint64_t *rowp = comb(xs - k);
                for (uint32_t p = 0; p <= xs - k; ++p) {
                    const uint64_t cp = ckl * rowp[p];
                    const double g = sum_s[xs - k - p] * pss[p];
                    const uint64_t *row = comb(xd - l);
                    for (uint32_t q = 0; q <= xd - l; ++q) {
                        const uint64_t c = cp * row[q];
                        inner += static_cast<double>(c) * ehalf[xs + xd - k - l - p - q] * g
                                * sum_d[xd - l - q] * hh[p + q] * psd[q];
                    }
                }
                prob += f * inner;
            }
        }
        return prob;
    }

    // :117-141: k, l = loci where the genotypes truly differ ...; the four binomials are one uint64_t
    // product, evaluated left to right as in the reference expression.
    double log_diff(uint32_t xs, uint32_t xd) const {
        double prob = diff_part(xs, xd, 0, xs + 1);
        prob *= static_cast<double>(comb(xs + xd)[xs]);
        return std::log(prob);
    }
};

}  // namespace

double reference_llr(double eps, double h, double theta, uint32_t x_s, uint32_t x_d) {
    if (x_s + x_d > kLlrRefMax) return std::nan("");
    const RefTables rt(eps, h, theta);
    return rt.log_diff(x_s, x_d) - rt.log_same(x_s, x_d);
}

// Beyond the table (read pairs sharing more than kLlrRefMax loci): one entry at a time, as the reference would
// evaluate it on first use (it memoises, :119, :155) -- O(x_s^2 x_d^2) terms, the outer index shared among up to
// max_threads threads (positive terms: the order of the partial sums moves the result by ulps). The tables are
// kept per process and rate triple and grow with the largest x_s + x_d asked for.
namespace {
std::mutex g_any_mutex;
struct AnyCache {
    std::unique_ptr<RefTables> tables;
    std::map<std::pair<uint32_t, uint32_t>, double> value;
};
std::map<std::array<double, 3>, AnyCache> &g_any_cache = *new std::map<std::array<double, 3>, AnyCache>();
}  // namespace

double reference_llr_any(double eps, double h, double theta, uint32_t x_s, uint32_t x_d, unsigned max_threads) {
    if (x_s + x_d <= kLlrRefMax) return reference_llr(eps, h, theta, x_s, x_d);
    std::lock_guard<std::mutex> lock(g_any_mutex);
    AnyCache &c = g_any_cache[{eps, h, theta}];
    const auto key = std::make_pair(x_s, x_d);
    const auto hit = c.value.find(key);
    if (hit != c.value.end()) return hit->second;
    if (!c.tables || c.tables->rows < x_s + x_d + 1) {
        uint32_t rows = 256;
        while (rows < x_s + x_d + 1) rows *= 2;
        c.tables.reset(new RefTables(eps, h, theta, rows));
    }
    const RefTables &rt = *c.tables;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_threads = std::max(1u, std::min({hw ? hw : 1u, 16u, std::max(1u, max_threads), x_s + 1}));
    std::vector<double> part(n_threads, 0.0);
    std::atomic<uint32_t> next{0};
    std::vector<double> per_k(x_s + 1, 0.0);
    auto work = [&]() {
        for (uint32_t k; (k = next.fetch_add(1)) <= x_s;) per_k[k] = rt.diff_part(x_s, x_d, k, k + 1);
    };
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < n_threads; ++i) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    double prob = 0;
    for (uint32_t k = 0; k <= x_s; ++k) prob += per_k[k];  // in the reference's order of k
    prob *= static_cast<double>(rt.comb(x_s + x_d)[x_s]);
    const double v = std::log(prob) - rt.log_same(x_s, x_d);
    c.value[key] = v;
    return v;
}

bool llr_exact_mode() {
    const char *env = std::getenv("SECEDO_LLR_EXACT");
    return env && std::atoi(env) != 0;
}

namespace {
// The reference-identical entries are a function of the three rates only and cost O(x_s^2 x_d^2) each
// (0.6 s for the whole triangle): computed once per process and rate triple, whatever the number of handles.
struct RefCacheEntry {
    uint32_t upto = 0;
    std::vector<double> value;  // kLlrTableDim^2; entries with 1 <= x_s + x_d <= upto are valid
};
std::mutex g_ref_mutex;
std::map<std::array<double, 3>, RefCacheEntry> &g_ref_cache = *new std::map<std::array<double, 3>, RefCacheEntry>();
}  // namespace

bool extend_reference(LlrTable *t, uint32_t max_shared, unsigned max_threads) {
    const uint32_t want = std::min(max_shared, kLlrRefMax);
    bool finite = true;
    auto check = [&](uint32_t upto) {
        for (uint32_t s = 0; s <= upto; ++s)
            for (uint32_t d = 0; s + d <= upto; ++d)
                if (s + d > 0 && !std::isfinite(t->value[s * kLlrTableDim + d])) finite = false;
    };
    if (llr_exact_mode() || want <= t->ref_upto) {
        check(want);
        return finite;
    }
    std::lock_guard<std::mutex> lock(g_ref_mutex);
    RefCacheEntry &cached = g_ref_cache[{t->eps, t->h, t->theta}];
    if (cached.value.empty()) cached.value.assign(kLlrTableDim * kLlrTableDim, 0.0);
    auto take = [&](uint32_t from, uint32_t upto) {
        for (uint32_t n = from + 1; n <= upto; ++n)
            for (uint32_t s = 0; s <= n; ++s) t->value[s * kLlrTableDim + (n - s)] = cached.value[s * kLlrTableDim + (n - s)];
    };
    // the entries (cached.upto, want], heaviest first, shared among a few threads (O(x_s^2 x_d^2) each:
    // about 1e8 terms for the whole triangle up to 64, 6e9 up to 128: two seconds on eight threads, once per process
    // and rate triple, and only for a pileup with a read of that many kept entries)
    std::vector<std::pair<uint32_t, uint32_t>> todo;
    for (uint32_t n = want; n > cached.upto; --n)
        for (uint32_t s = 0; s <= n; ++s) todo.push_back({s, n - s});
    auto cost = [](const std::pair<uint32_t, uint32_t> &e) {
        return (uint64_t)(e.first + 1) * (e.first + 1) * (e.second + 1) * (e.second + 1);
    };
    std::sort(todo.begin(), todo.end(), [&](const auto &a, const auto &b) { return cost(a) > cost(b); });
    const RefTables rt(t->eps, t->h, t->theta);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < todo.size();) {
            const uint32_t s = todo[i].first, d = todo[i].second;
            cached.value[s * kLlrTableDim + d] = rt.log_diff(s, d) - rt.log_same(s, d);
        }
    };
    if (!todo.empty()) {
        // helper threads only for the heavy part of the triangle, and no more than the caller's num_threads
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned n_threads = want >= 32 ? std::max(1u, std::min({hw ? hw : 1u, 16u, std::max(1u, max_threads)})) : 1u;
        std::vector<std::thread> pool;
        for (unsigned i = 1; i < n_threads; ++i) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
        cached.upto = want;
    }
    take(t->ref_upto, want);
    t->ref_upto = want;
    double per_locus = 0;
    for (uint32_t s = 0; s < kLlrTableDim; ++s)
        for (uint32_t d = 0; d < kLlrTableDim; ++d) {
            const double v = t->value[s * kLlrTableDim + d];
            if (s + d && std::isfinite(v)) per_locus = std::max(per_locus, std::fabs(v) / (s + d));
        }
    t->max_abs_per_locus = per_locus;
    check(want);
    return finite;
}

LlrTable make_llr_table(double eps, double h, double theta, uint64_t pair_bound) {
    LlrTable t;
    t.eps = eps;
    t.h = h;
    t.theta = theta;
    t.model = make_llr_model(eps, h, theta);
    t.value.assign(kLlrTableDim * kLlrTableDim, 0.0);
    t.fixed.assign(kLlrTableDim * kLlrTableDim, 0);
    double per_locus = 0;
    for (uint32_t s = 0; s < kLlrTableDim; ++s) {
        for (uint32_t d = 0; d < kLlrTableDim; ++d) {
            if (s + d == 0) continue;
            const double v = llr(t.model, s, d);
            t.value[s * kLlrTableDim + d] = v;
            if (std::isfinite(v)) per_locus = std::max(per_locus, std::fabs(v) / (s + d));
        }
    }
    t.max_abs_per_locus = per_locus;
    requantize(&t, llr_scale_for(t, pair_bound, kLlrRefMax + 1));
    return t;
}

double llr_per_locus_bound(const LlrTable &t, uint32_t max_shared) {
    // No read pair shares more loci than its shorter read has kept entries, so only the entries with
    // x_s + x_d <= max_shared can be added; a pair that shares n loci is n incidences and adds |D| <= n * (the
    // largest |D| / n among them). The reach is rounded up to a power of two so that the packing paths, whose
    // longest-read figures may differ (an upper bound on the device, the exact one on the host), agree on it.
    uint32_t reach = 1;
    while (reach < max_shared && reach < kLlrRefMax) reach *= 2;
    const bool whole = max_shared > kLlrRefMax;  // beyond the table D grows linearly: the table's maximum, with margin
    double per_locus = 0;
    for (uint32_t s = 0; s < kLlrTableDim; ++s)
        for (uint32_t d = 0; d < kLlrTableDim; ++d) {
            if (s + d == 0 || (!whole && s + d > reach)) continue;
            const double v = t.value[s * kLlrTableDim + d];
            if (std::isfinite(v)) per_locus = std::max(per_locus, std::fabs(v) / (s + d));
        }
    return whole ? std::max(1.0, 1.5 * per_locus) : per_locus;
}

int llr_scale_for(const LlrTable &t, uint64_t pair_bound, uint32_t max_shared) {
    // |sum over one cell pair| <= per_locus * incidences (1.5x margin: rounding of the terms, the additions of
    // the count tile's conversion); keep it below 2^62. Partial sums may wrap on the way -- two's complement
    // addition is exact modulo 2^64 -- only the final sum has to fit.
    const double bound = 1.5 * llr_per_locus_bound(t, max_shared) * static_cast<double>(std::max<uint64_t>(pair_bound, 1));
    int k = 44;
    while (k > 0 && std::ldexp(bound, k) >= std::ldexp(1.0, 62)) --k;
    return k;
}

void requantize(LlrTable *t, int k) {
    t->scale_log2 = k;
    t->fixed.assign(t->value.size(), 0);
    for (size_t i = 0; i < t->value.size(); ++i) {
        const double v = t->value[i];
        t->fixed[i] = std::isfinite(v) ? static_cast<int64_t>(std::llround(std::ldexp(v, k))) : 0;
    }
}

}  // namespace secedo
