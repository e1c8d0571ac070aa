// sym_eig.hpp -- host eigen-decomposition of a small dense symmetric matrix (see sym_eig.cpp)
#pragma once

#include <vector>

namespace secedo {

// a: n x n row-major (symmetrised as (a + a^T) / 2). evals ascending; evecs n x n row-major with the
// eigenvector of evals[k] in COLUMN k. Returns false if the QL iteration did not converge.
bool sym_eig(int n, const std::vector<double> &a, std::vector<double> &evals, std::vector<double> &evecs);

// All eigenvalues (ascending) but only the eigenvectors of the k LARGEST: top_vecs is n x k row-major,
// column j = eigenvector of evals[n - 1 - j]. Householder reduction with the reflectors kept, values by
// QL without accumulation, vectors by inverse iteration on the tridiagonal and back-transformation:
// about a fifth of the work of sym_eig for k = 32 of 192. Falls back to sym_eig if a vector does not
// verify against the matrix.
// top_values_only: only evals[n - k .. n - 1] are computed (bisection instead of QL; the rest is NaN).
bool sym_eig_top(int n, const std::vector<double> &a, int k, std::vector<double> &evals, std::vector<double> &top_vecs,
                 bool top_values_only = false);

}  // namespace secedo
