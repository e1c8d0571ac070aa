// sym_eig.hpp -- host eigen-decomposition of a small dense symmetric matrix (see sym_eig.cpp)
#pragma once

#include <vector>

namespace secedo {

// a: n x n row-major (symmetrised as (a + a^T) / 2). evals ascending; evecs n x n row-major with the
// eigenvector of evals[k] in COLUMN k. Returns false if the QL iteration did not converge.
bool sym_eig(int n, const std::vector<double> &a, std::vector<double> &evals, std::vector<double> &evecs);

}  // namespace secedo
