// llr_table.hpp -- log-likelihood-ratio terms of the similarity-matrix path (host side).
//
// The reference evaluates log P(x_s,x_d | same genotype) and log P(x_s,x_d | different) by
// nested binomial sums (reference: similarity_matrix.cpp:153-170 and :117-141) over tables of
// powers and a u64 Pascal triangle (struct Cache, :38-104). Both sums are binomial/trinomial
// expansions and collapse to closed forms (derivation in DESIGN.md, "LLR closed form"):
//
//   P_same(x_s,x_d) = C(n,x_s)/2 * ( pss^x_s psd^x_d + w2^x_s z2^x_d )
//   P_diff(x_s,x_d) = C(n,x_s)/2 * ( u1^x_s v1^x_d + u2^x_s v2^x_d ),   n = x_s + x_d
//
// so D = log P_diff - log P_same needs no binomial at all. Only D leaves the reference function
// (similarity_matrix.cpp:428), hence only D is tabulated here.
//
// The closed form is the value of the reference's formula in exact arithmetic. The reference itself
// multiplies its binomials as uint64_t before the product becomes a double (:125, :159), and those
// products wrap from about x_s + x_d = 48 on: what it returns there is the wrapped sum, not the formula
// (D(60,4) differs by 0.11). A drop-in must return what the reference returns, so the table entries a
// pileup can reach (x_s + x_d <= longest read, up to 128; 64 until round 3) are evaluated by reference_llr(),
// which restates the reference's nested sums with the same wrapping integer arithmetic (from row 68 on the
// single binomials of its uint64 Pascal triangle wrap as well: the same modular additions here); the closed
// form serves beyond the table (read pairs sharing more than 128 loci) and under SECEDO_LLR_EXACT=1.
#pragma once

#include <cstdint>
#include <vector>

namespace secedo {

// Natural logs of the eight bases of the closed form; what the device needs to evaluate D for
// (x_s, x_d) outside the table.
struct LlrModel {
    double ln_u1, ln_v1, ln_u2, ln_v2;  // different-genotype mixture components
    double ln_w1, ln_z1, ln_w2, ln_z2;  // same-genotype mixture components
};

LlrModel make_llr_model(double mutation_rate, double homozygous_rate, double seq_error_rate);

// D(x_s, x_d), evaluated in log space (stable for any x_s, x_d).
double llr(const LlrModel &m, uint32_t x_s, uint32_t x_d);

constexpr uint32_t kLlrTableDim = 129;  // x_s, x_d in [0, 128]
constexpr uint32_t kLlrRefMax = 128;    // entries with x_s + x_d <= this can be made reference-identical

// D(x_s, x_d) as the reference computes it: log of the four-fold sum of :117-141 minus log of the
// two-fold sum of :153-170, binomials multiplied in uint64_t (wrap-around kept). x_s + x_d <= kLlrRefMax.
double reference_llr(double mutation_rate, double homozygous_rate, double seq_error_rate, uint32_t x_s,
                     uint32_t x_d);

// The same for any x_s + x_d (tables as long as needed; the reference's go up to max_fragment_length): what the
// reference returns for a read pair that shares more than kLlrRefMax loci -- by then an artefact of its wrapped
// integer arithmetic, reproduced all the same. Cached per process, rate triple and (x_s, x_d).
double reference_llr_any(double mutation_rate, double homozygous_rate, double seq_error_rate, uint32_t x_s,
                         uint32_t x_d, unsigned max_threads = 1);

struct LlrTable {
    LlrModel model;
    double eps = 0, h = 0, theta = 0;
    uint32_t ref_upto = 0;         // entries with 1 <= x_s + x_d <= ref_upto hold reference_llr()
    int scale_log2;                // fixed point: value = round(D * 2^scale_log2)
    double max_abs_per_locus;      // max over the table of |D| / (x_s + x_d)
    std::vector<int64_t> fixed;    // kLlrTableDim^2, row = x_s
    std::vector<double> value;     // same, in double
};

// pair_bound: upper bound on the number of (read pair, shared locus) incidences that can land on
// one cell pair; picks the largest scale (<= 44) for which the int64 accumulator cannot overflow.
LlrTable make_llr_table(double mutation_rate, double homozygous_rate, double seq_error_rate,
                        uint64_t pair_bound);

// largest |D(x_s, x_d)| / (x_s + x_d) over the entries a pileup whose longest read has max_shared kept
// entries can reach
double llr_per_locus_bound(const LlrTable &t, uint32_t max_shared);
// largest fixed-point scale (<= 44) for which an int64 accumulator cannot overflow: pair_bound bounds the
// (read pair, shared locus) incidences of one cell pair, max_shared the loci one read pair can share
int llr_scale_for(const LlrTable &t, uint64_t pair_bound, uint32_t max_shared);
// re-derive t->fixed from t->value for another scale
void requantize(LlrTable *t, int scale_log2);
// Make the entries with x_s + x_d <= max_shared (capped at kLlrRefMax) reference-identical (no-op for
// those that already are, and under SECEDO_LLR_EXACT=1). t->fixed is stale afterwards: requantize.
// Returns false when one of those entries is not finite (rates for which the reference itself
// produces inf / NaN, e.g. a sequencing error rate of 0).
// Computed once per process and rate triple; at most max_threads helper threads (the caller's num_threads).
bool extend_reference(LlrTable *t, uint32_t max_shared, unsigned max_threads = 1);
bool llr_exact_mode();  // SECEDO_LLR_EXACT=1: the closed form everywhere

}  // namespace secedo
