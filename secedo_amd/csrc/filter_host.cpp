#include "filter_host.hpp"

#include <algorithm>
#include <cfenv>
#include <cmath>

namespace secedo {

// numeric constants of the algorithm (util/is_significant.cpp:11-47; produced upstream by
// scripts/K.r): optimal thresholds per coverage decile (columns: 10, 20, ..., 200) and expected
// split of the cells (rows: 10-90, 20-80, 30-70, 40-60, 50-50)
const double kSignificanceThresholds[5][20] = {
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

int is_significant(const uint16_t base_count[4], double theta, uint32_t cell_proportion,
                   double *statistic, double *threshold) {
    if (statistic) *statistic = std::nan("");
    if (threshold) *threshold = std::nan("");
    if (cell_proportion > 4) return 0;
    uint16_t c[4] = {base_count[0], base_count[1], base_count[2], base_count[3]};
    const uint32_t coverage = static_cast<uint32_t>(c[0]) + c[1] + c[2] + c[3];
    if (coverage < 2) return 0;              // no reads, or a single one (:83-85)
    std::sort(c, c + 4);                     // ascending (:88)
    if (c[2] == 0) return 0;                 // all bases equal (:90-92)
    if (c[2] + c[1] + c[0] < 5) return 0;    // fewer than 5 deviating bases (:97-99)
    if (c[3] < 1.5 * c[2]) return 0;         // no clear majority base (:101-103)

    std::fesetround(FE_TONEAREST);
    const double col = std::clamp(std::nearbyint(coverage / 10.) - 1, 0., 19.);  // :106-107
    const double k = kSignificanceThresholds[cell_proportion][static_cast<uint32_t>(col)];

    const double hetero_prior = 0.0005, mut_prior = 1e-6;       // :52-57
    const double homo_prior = 1 - hetero_prior - mut_prior;
    // log P(counts | homozygous) with the two priors the reference adds (:110-116; the second one is
    // log(hetero_prior) in the reference, kept)
    double log_homozygous = c[3] * std::log(1 - theta) + (coverage - c[3]) * std::log(theta / 3);
    log_homozygous += std::log(1. / 4);
    log_homozygous += std::log(hetero_prior);
    // evidence: five hypotheses (:120-135)
    const double all_c1 = homo_prior * std::pow(1 - theta, c[3]) * std::pow(theta / 3, coverage - c[3]);
    const double hetero = hetero_prior * std::pow(0.5 - theta / 3, c[3] + c[2]) * std::pow(theta / 3, c[0] + c[1]);
    const double homo_som = homo_prior * mut_prior * std::pow(0.75 - 2 * theta / 3, c[3]) * std::pow(0.25, c[2])
            * std::pow(theta / 3, c[0] + c[1]);
    const double hetero_som = hetero_prior * mut_prior * std::pow(0.5 - theta, c[3]) * std::pow(0.25, c[1] + c[2])
            * std::pow(theta / 3, c[0]);
    const double two_som = hetero_prior * mut_prior * mut_prior * std::pow(1 - theta, coverage);
    const double s = log_homozygous - std::log(all_c1 + hetero + homo_som + hetero_som + two_som);
    if (statistic) *statistic = s;
    if (threshold) *threshold = k;
    return s < k ? 1 : 0;  // :137
}

}  // namespace secedo
