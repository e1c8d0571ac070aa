/*
 * secedo_spectral.h -- C-ABI of the spectral step that consumes the similarity matrix
 * (SURVEY.md section 8f, rank 1).
 *
 * Replaces, in the reference's spectral_clustering():
 *     Matd L = laplacian(similarity);                     (spectral_clustering.cpp:33-52, :127)
 *     arma::eig_sym(eigenvalues, eigenvectors, lap);      (spectral_clustering.cpp:136-138)
 * of which the reference uses the 20 smallest eigenvalues (:141-143, a CSV for inspection) and the
 * eigenvectors of the 7 smallest (columns 0..6: :166, :171-172, :221, :235-237). The dense O(N^3)
 * decomposition is replaced by a block Krylov iteration on the matrix where it already lies (HBM):
 * the normalised Laplacian is never formed, the GMM / k-means that follow stay on the host.
 *
 * L = I - D^-1/2 A D^-1/2 with D = diag(row sums of A), 1/sqrt(0) := 0 as in the reference (:40-41).
 * Eigenvalues ascending (eig_sym order). An eigenvector's sign is arbitrary in LAPACK; here the
 * component of largest magnitude (lowest index on ties) is made positive.
 *
 * Error codes and secedo_simmat_last_error() are those of secedo_simmat.h. No CPU fallback.
 */
#ifndef SECEDO_SPECTRAL_H
#define SECEDO_SPECTRAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SECEDO_SPECTRAL_MAX_VALUES 32u

typedef struct secedo_spectral_info {
    uint32_t cycles;        /* restart cycles run */
    uint32_t block_products; /* products of the operator with a block of 32 vectors */
    uint32_t converged;     /* 1 when both residual bounds below were met */
    uint32_t reserved;
    double max_residual_vectors; /* max ||L v - lambda v|| over the n_vectors returned pairs */
    double max_residual_values;  /* ... over all n_values pairs */
} secedo_spectral_info;

/* laplacian() (spectral_clustering.cpp:33-52), dense, device to device: d_out[n*n] row-major.
 * d_similarity must be symmetric with a zero diagonal (the reference asserts both, :36,38). */
int secedo_laplacian_device(const double *d_similarity, uint32_t n, double *d_out, void *stream);

/* The n_values smallest eigenvalues of the normalised Laplacian of d_similarity (device, n*n
 * row-major, symmetric, zero diagonal) and the eigenvectors of the n_vectors smallest.
 *   eigenvalues      host, n_values doubles, ascending
 *   d_eigenvectors   device, column-major n x n_vectors (column j = eigenvector j, the layout of
 *                    arma::mat), unit 2-norm; may be NULL when n_vectors == 0
 *   tol              residual bound ||L v - lambda v|| for the returned eigenvectors (0 -> 1e-9);
 *                    the eigenvalues beyond n_vectors are iterated to sqrt-ish accuracy 1e-6, which
 *                    bounds their error by 1e-12 / gap
 *   max_cycles       restart cycles before giving up (0 -> 60); not converging is not an error:
 *                    info->converged says so and the best pairs found are returned
 * 1 <= n_vectors <= n_values <= min(n, SECEDO_SPECTRAL_MAX_VALUES), or n_vectors == 0. */
int secedo_spectral_eigs_device(int device_id, const double *d_similarity, uint32_t n, uint32_t n_values,
                                uint32_t n_vectors, double tol, uint32_t max_cycles, double *eigenvalues,
                                double *d_eigenvectors, secedo_spectral_info *info, void *stream);

/* The matrix sharded by rows over several ranks (one process per GPU; BASELINE config 5: the matrix
 * is never gathered). Every rank passes the rows [row_begin, row_begin + n_rows) it holds
 * (d_rows[n_rows * n], e.g. from secedo_simmat_finalize_rows; the row blocks of the ranks partition
 * [0, n), a rank may hold none) and a callback that sums a device buffer of doubles over all ranks in
 * place, on `stream` (RCCL ncclAllReduce, torch.distributed.all_reduce, ...; 0 on success). Per product
 * with a block of 32 vectors each rank multiplies its rows and the n x 32 partial results are summed:
 * 8 N^2 / ranks bytes of matrix per rank against one all-reduce of 256 N bytes. Everything else is
 * replicated and deterministic, so every rank returns the same eigenvalues and the full eigenvectors.
 * allreduce == NULL requires the full matrix (row_begin 0, n_rows n). */
typedef int (*secedo_allreduce_sum_fn)(void *ctx, double *d_buffer, uint64_t count, void *stream);
int secedo_spectral_eigs_rows_device(int device_id, const double *d_rows, uint32_t row_begin, uint32_t n_rows,
                                     uint32_t n, uint32_t n_values, uint32_t n_vectors, double tol,
                                     uint32_t max_cycles, double *eigenvalues, double *d_eigenvectors,
                                     secedo_spectral_info *info, secedo_allreduce_sum_fn allreduce,
                                     void *allreduce_ctx, void *stream);

/* Same with host buffers (similarity n*n row-major in, eigenvectors column-major out): what a
 * caller holding a Matd uses. */
int secedo_spectral_eigs(int device_id, const double *similarity, uint32_t n, uint32_t n_values,
                         uint32_t n_vectors, double tol, uint32_t max_cycles, double *eigenvalues,
                         double *eigenvectors, secedo_spectral_info *info);

#ifdef __cplusplus
}
#endif

#endif /* SECEDO_SPECTRAL_H */
