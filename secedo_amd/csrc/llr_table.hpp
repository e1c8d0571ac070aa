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

constexpr uint32_t kLlrTableDim = 65;  // x_s, x_d in [0, 64]: one 32-locus window either side + 1

struct LlrTable {
    LlrModel model;
    int scale_log2;                // fixed point: value = round(D * 2^scale_log2)
    double max_abs_per_locus;      // max over the table of |D| / (x_s + x_d)
    std::vector<int64_t> fixed;    // kLlrTableDim^2, row = x_s
    std::vector<double> value;     // same, in double
};

// pair_bound: upper bound on the number of (read pair, shared locus) incidences that can land on
// one cell pair; picks the largest scale (<= 44) for which the int64 accumulator cannot overflow.
LlrTable make_llr_table(double mutation_rate, double homozygous_rate, double seq_error_rate,
                        uint64_t pair_bound);

// largest fixed-point scale (<= 44) for which an int64 accumulator cannot overflow given the bound
int llr_scale_for(const LlrTable &t, uint64_t pair_bound);
// re-derive t->fixed from t->value for another scale
void requantize(LlrTable *t, int scale_log2);

}  // namespace secedo
