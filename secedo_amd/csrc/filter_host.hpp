// filter_host.hpp -- host side of the locus filter: the significance test on base counts.
// (reference: util/is_significant.cpp:78-138 Filter::is_significant; thresholds :11-47)
#pragma once

#include <cstdint>

namespace secedo {

constexpr uint32_t kNoPos = 16383;  // util/is_significant.hpp:11 (NO_POS)

// 1 if the locus is kept. `statistic`/`threshold` (optional) receive the two sides of the final
// comparison, NaN when an integer pre-test already rejected the locus.
int is_significant(const uint16_t base_count[4], double theta, uint32_t cell_proportion,
                   double *statistic = nullptr, double *threshold = nullptr);

// thresholds K[cell_proportion][min(19, max(0, round_half_even(coverage / 10) - 1))]
extern const double kSignificanceThresholds[5][20];

}  // namespace secedo
