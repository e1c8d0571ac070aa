// spectral_kernels.hpp -- launch wrappers of spectral_kernels.hip (gfx950). See spectral_api.cpp for
// the iteration they serve. Block vectors are row-major n x 32 doubles ("32 interleaved vectors").
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace secedo {
namespace spectral {

constexpr uint32_t kBlockWidth = 32;  // vectors per block: two 16-wide MFMA tiles
constexpr uint32_t kGramChunk = 128;  // rows per workgroup of the Gram kernel

// sums[row_begin + r] = sum_j A_rows[r][j] for the n_rows local rows (the rest of sums is left alone)
hipError_t row_sums(const double *A_rows, uint32_t n, uint32_t row_begin, uint32_t n_rows, double *sums,
                    hipStream_t stream);
// s[i] = 1 / sqrt(sums[i]) (0 when the sum is 0; reference spectral_clustering.cpp:34-43), root[i] = sqrt(sums[i])
hipError_t scale_from_sums(uint32_t n, const double *sums, double *s, double *root, hipStream_t stream);

// out = I - diag(s) A diag(s)   (spectral_clustering.cpp:44-50)
hipError_t laplacian(const double *A, const double *s, uint32_t n, double *out, hipStream_t stream);

// X[:, 0] = root (the known eigenvector of eigenvalue 0), the other columns a fixed pseudo-random fill
hipError_t init_block(uint32_t n, const double *root, double *X, hipStream_t stream);

// number of row segments the product over n_rows local rows is split into; padded row counts
uint32_t product_segments(uint32_t n, uint32_t n_rows);
inline uint32_t pad16(uint32_t n) { return (n + 15u) / 16u * 16u; }

// T = (I + diag(s) A diag(s)) / 2 applied to a block X in two halves, so that ranks holding row blocks
// of the symmetric A can add their parts in between:
//   product_partial  Ypart = sum over the local rows j of (s[j] X[j]) A[j][:]      (n x 32)
//   product_finish   Y = (X + s o Ysum) / 2, Ysum = the sum of all ranks' Ypart
// Z ((pad16(n) + 64) x 32) and P (segments x pad16(n) x 32) are scratch. A single rank passes Y_finished
// (and no Ypart): the last kernel of product_partial then finishes the product itself.
hipError_t product_partial(const double *A_rows, uint32_t n, uint32_t row_begin, uint32_t n_rows, const double *s,
                           const double *X, double *Z, double *P, double *Ypart, double *Y_finished,
                           hipStream_t stream);
hipError_t product_finish(uint32_t n, const double *s, const double *X, const double *Ysum, double *Y,
                          hipStream_t stream);

// G[blk] (32 x 32) = Q[blk]^T W for blk < nblk; Q blocks blk_stride doubles apart; Gp scratch of
// gram_chunks(n) * nblk * 1024 doubles
uint32_t gram_chunks(uint32_t n);
hipError_t gram(uint32_t n, const double *Q, size_t blk_stride, uint32_t nblk, const double *W, double *Gp,
                double *G, hipStream_t stream);

// out = beta * out + alpha * sum_blk Q[blk] M[blk]   (M[blk] 32 x 32 row-major); out must not alias Q
hipError_t block_combine(uint32_t n, const double *Q, size_t blk_stride, uint32_t nblk, const double *M,
                         double alpha, double beta, double *out, hipStream_t stream);

// Cholesky factor R (upper triangular, row-major 32 x 32) of a Gram matrix G = W^T W with column dropping
// (see k_cholesky_drop), Rinv = its inverse on the surviving columns, alive[32] = who survived. With
// R_prev the R that is written is R * R_prev (the factor of two passes together). All on the device.
hipError_t cholesky_drop(const double *G, const double *R_prev, double *Rinv, double *R, uint32_t *alive,
                         hipStream_t stream);

// out (column-major n x k) = the first k columns of Y, each scaled to unit norm and signed so that
// its component of largest magnitude (lowest index on ties) is positive
hipError_t write_vectors(uint32_t n, const double *Y, uint32_t k, double *out, hipStream_t stream);

}  // namespace spectral
}  // namespace secedo
