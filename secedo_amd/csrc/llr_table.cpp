#include "llr_table.hpp"

#include <algorithm>
#include <cmath>

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

LlrTable make_llr_table(double eps, double h, double theta, uint64_t pair_bound) {
    LlrTable t;
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
    requantize(&t, llr_scale_for(t, pair_bound));
    return t;
}

int llr_scale_for(const LlrTable &t, uint64_t pair_bound) {
    // |sum over one cell pair| <= per_locus * incidences; keep it below 2^62 (1.5x margin for the
    // terms beyond the table, which grow linearly in x_s + x_d as well)
    const double bound = std::max(1.0, 1.5 * t.max_abs_per_locus) * static_cast<double>(std::max<uint64_t>(pair_bound, 1));
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
