// spectral_api.cpp -- host side of include/secedo_spectral.h.
//
// Smallest eigenpairs of L = I - D^-1/2 A D^-1/2 (reference: laplacian() + arma::eig_sym,
// spectral_clustering.cpp:33-52, :127-138) without forming L and without the O(N^3) decomposition.
// The wanted pairs are the LARGEST of T = (I + D^-1/2 A D^-1/2) / 2 (spectrum in [0, 1],
// lambda_L = 2 (1 - tau)), found by a restarted block Lanczos iteration:
//   * blocks of 32 vectors: one pass over the N x N matrix serves 32 vectors (the block product runs on
//     the fp64 matrix cores and is bound by the 8 N^2 bytes of A);
//   * full re-orthogonalisation against the cycle's basis (classical Gram-Schmidt twice) and a
//     Cholesky QR of each new block, with dependent columns dropped -- so eigenvalue multiplicity
//     up to 32 (disconnected cell graphs) and N < 32 need no special case;
//   * Rayleigh-Ritz on the (6 x 32)-dimensional projection every cycle, on the host (sym_eig.cpp: all
//     Ritz values, the 32 Ritz vectors that are kept);
//     residuals from the last coupling block; restart from the 32 best Ritz vectors.
// The first start vector is D^1/2 1, the known eigenvector of eigenvalue 0.
// The cell-cluster eigenvectors converge in the first cycle or two. The rest of the 20 values the
// reference logs (and of the 7 vectors it uses when there are fewer clusters than that) sit at the
// dense edge of the bulk of the spectrum, where the residual halves per cycle; a Chebyshev filter on
// the bulk was tried and only pays from degree 32 on, where its products cost more than the cycles
// they save (N = 8000: 0.36 s against 0.18 s) -- it is not in the code.
#include "secedo_simmat.h"
#include "secedo_spectral.h"
#include "spectral_kernels.hpp"
#include "sym_eig.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace secedo {
int api_fail(int code, const std::string &msg);  // simmat_api.cpp: sets secedo_simmat_last_error()
}

namespace {

using secedo::spectral::kBlockWidth;
constexpr uint32_t BW = kBlockWidth;
// Krylov blocks per restart cycle, set per solve (cycle_blocks): a larger space needs fewer matrix passes in
// all (N = 8000: 132 / 112 / 104 block products with 6 / 7 / 8 blocks) but its Rayleigh-Ritz problem on the host
// grows with the cube -- measured best: 6 below 2000 rows (17.8 / 19.5 / 22.8 ms at N = 1000), 7 up to 12000
// (70.6 / 60.6 / 63.5 ms at N = 8000), 8 beyond (137 / 127 / 118 ms at N = 16000).
thread_local uint32_t kCycleBlocks = 6;
uint32_t cycle_blocks(uint32_t n) {
    if (const char *e = std::getenv("SECEDO_SPECTRAL_BLOCKS")) {
        const int v = std::atoi(e);
        if (v >= 3 && v <= 12) return (uint32_t)v;
    }
    return n < 2000u ? 6u : n < 12000u ? 7u : 8u;
}

struct Buf {
    void *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    double *d() const { return static_cast<double *>(p); }
};

#define SP_TRY(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return secedo::api_fail(SECEDO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

struct Solver {
    uint32_t n = 0, row_begin = 0, n_rows = 0;
    hipStream_t stream = nullptr;
    const double *A = nullptr;  // rows [row_begin, row_begin + n_rows) of the matrix
    secedo_allreduce_sum_fn allreduce = nullptr;
    void *allreduce_ctx = nullptr;
    Buf s, root, sums, Q, W, W2, Wtmp, Z, P, Ypart, Gp, G, Gall, M, Rfirst, Rblk, alive_dev;
    size_t blk_stride = 0;

    int setup(const double *d_rows, uint32_t row_begin_, uint32_t n_rows_, uint32_t n_, secedo_allreduce_sum_fn fn,
              void *ctx, hipStream_t st) {
        using namespace secedo::spectral;
        n = n_;
        row_begin = row_begin_;
        n_rows = n_rows_;
        stream = st;
        A = d_rows;
        allreduce = fn;
        allreduce_ctx = ctx;
        blk_stride = (size_t)n * BW;
        SP_TRY(sums.alloc((size_t)n * 8));
        SP_TRY(Ypart.alloc(blk_stride * 8));
        SP_TRY(s.alloc((size_t)n * 8));
        SP_TRY(root.alloc((size_t)n * 8));
        SP_TRY(Q.alloc((kCycleBlocks + 1) * blk_stride * 8));
        SP_TRY(W.alloc(blk_stride * 8));
        SP_TRY(W2.alloc(blk_stride * 8));
        SP_TRY(Wtmp.alloc(blk_stride * 8));
        SP_TRY(Z.alloc(((size_t)pad16(n) + 64) * BW * 8));
        SP_TRY(P.alloc((size_t)product_segments(n, n_rows) * pad16(n) * BW * 8));
        SP_TRY(Gp.alloc((size_t)gram_chunks(n) * (kCycleBlocks + 1) * BW * BW * 8));
        SP_TRY(G.alloc((size_t)(kCycleBlocks + 1) * BW * BW * 8));
        SP_TRY(M.alloc((size_t)(kCycleBlocks + 1) * BW * BW * 8));
        // Gram-Schmidt coefficients of a whole cycle (6 steps x 2 passes x up to 6 blocks): read back once
        SP_TRY(Gall.alloc((size_t)kCycleBlocks * 2 * kCycleBlocks * BW * BW * 8));
        SP_TRY(Rfirst.alloc((size_t)BW * BW * 8));
        SP_TRY(Rblk.alloc((size_t)(kCycleBlocks + 1) * BW * BW * 8));
        SP_TRY(alive_dev.alloc((size_t)(kCycleBlocks + 1) * BW * 4));
        return SECEDO_OK;
    }
    double *block(uint32_t b) const { return Q.d() + b * blk_stride; }
    // sum over the ranks of a device buffer, in place (nothing to do for a rank that holds all rows)
    int reduce_ranks(double *buf, size_t count) {
        if (!allreduce) return SECEDO_OK;
        if (allreduce(allreduce_ctx, buf, count, stream) != 0)
            return secedo::api_fail(SECEDO_E_STATE, "the all-reduce callback of the spectral step failed");
        return SECEDO_OK;
    }
    // D^-1/2 from the row sums of the whole matrix
    int scales() {
        using namespace secedo::spectral;
        SP_TRY(hipMemsetAsync(sums.p, 0, (size_t)n * 8, stream));
        SP_TRY(row_sums(A, n, row_begin, n_rows, sums.d(), stream));
        const int rc = reduce_ranks(sums.d(), n);
        if (rc) return rc;
        SP_TRY(scale_from_sums(n, sums.d(), s.d(), root.d(), stream));
        return SECEDO_OK;
    }
    // y = T x
    int product(const double *x, double *y) {
        using namespace secedo::spectral;
        // (one rank: the segments' sum and the finishing step in one kernel)
        SP_TRY(product_partial(A, n, row_begin, n_rows, s.d(), x, Z.d(), P.d(), allreduce ? Ypart.d() : nullptr,
                               allreduce ? nullptr : y, stream));
        if (!allreduce) return SECEDO_OK;
        const int rc = reduce_ranks(Ypart.d(), blk_stride);
        if (rc) return rc;
        SP_TRY(product_finish(n, s.d(), x, Ypart.d(), y, stream));
        return SECEDO_OK;
    }
    int upload_small(const std::vector<double> &m) {
        SP_TRY(hipMemcpyAsync(M.d(), m.data(), m.size() * 8, hipMemcpyHostToDevice, stream));
        return SECEDO_OK;
    }
    // Q[blk] = orthonormalised src (Cholesky QR, twice, through Wtmp; src keeps its contents). Nothing comes back to the
    // host here: R with src = Q[blk] R goes to Rblk[blk] and the surviving columns to alive_dev[blk], which
    // the cycle reads once, with the Gram-Schmidt coefficients.
    int orthonormalise(double *src, uint32_t blk) {
        using namespace secedo::spectral;
        double *dst = block(blk);
        double *R = static_cast<double *>(Rblk.p) + (size_t)blk * BW * BW;
        uint32_t *alive = static_cast<uint32_t *>(alive_dev.p) + (size_t)blk * BW;
        SP_TRY(gram(n, src, blk_stride, 1, src, Gp.d(), G.d(), stream));
        SP_TRY(cholesky_drop(G.d(), nullptr, M.d(), Rfirst.d(), nullptr, stream));
        double *tmp = Wtmp.d();
        SP_TRY(block_combine(n, src, blk_stride, 1, M.d(), 1.0, 0.0, tmp, stream));
        SP_TRY(gram(n, tmp, blk_stride, 1, tmp, Gp.d(), G.d(), stream));
        SP_TRY(cholesky_drop(G.d(), Rfirst.d(), M.d(), R, alive, stream));
        SP_TRY(block_combine(n, tmp, blk_stride, 1, M.d(), 1.0, 0.0, dst, stream));
        return SECEDO_OK;
    }
};

int solve(int device_id, const double *d_rows, uint32_t row_begin, uint32_t n_rows, uint32_t n, uint32_t n_values,
          uint32_t n_vectors, double tol, uint32_t max_cycles, double *eigenvalues, double *d_eigenvectors,
          secedo_spectral_info *info, secedo_allreduce_sum_fn allreduce, void *allreduce_ctx, hipStream_t stream) {
    using namespace secedo::spectral;
    if ((!d_rows && n_rows) || !eigenvalues) return secedo::api_fail(SECEDO_E_INVALID_ARG, "null argument");
    if ((uint64_t)row_begin + n_rows > n) return secedo::api_fail(SECEDO_E_INVALID_ARG, "row block outside the matrix");
    if (!allreduce && n_rows != n)
        return secedo::api_fail(SECEDO_E_INVALID_ARG, "a row block needs the all-reduce callback of the other ranks");
    if (n == 0) return secedo::api_fail(SECEDO_E_INVALID_ARG, "the similarity matrix is empty");
    if (n_values == 0 || n_values > std::min<uint32_t>(n, SECEDO_SPECTRAL_MAX_VALUES))
        return secedo::api_fail(SECEDO_E_INVALID_ARG, "n_values must be in [1, min(n, 32)]");
    if (n_vectors > n_values) return secedo::api_fail(SECEDO_E_INVALID_ARG, "n_vectors must not exceed n_values");
    if (n_vectors && !d_eigenvectors) return secedo::api_fail(SECEDO_E_INVALID_ARG, "d_eigenvectors is null");
    if ((uint64_t)n * n >= (1ull << 40)) return secedo::api_fail(SECEDO_E_LIMIT, "matrix too large");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return secedo::api_fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the spectral step has no CPU fallback");
    if (device_id < 0 || device_id >= n_dev) return secedo::api_fail(SECEDO_E_NO_DEVICE, "device id out of range");
    SP_TRY(hipSetDevice(device_id));
    if (tol <= 0.0) tol = 1e-9;
    if (max_cycles == 0) max_cycles = 60;
    const double tol_values = std::max(tol, 1e-6);

    Solver sv;
    kCycleBlocks = cycle_blocks(n);
    int rc = sv.setup(d_rows, row_begin, n_rows, n, allreduce, allreduce_ctx, stream);
    if (rc) return rc;
    if ((rc = sv.scales())) return rc;
    SP_TRY(init_block(n, sv.root.d(), sv.W.d(), stream));
    std::vector<double> R_last((size_t)BW * BW);
    std::vector<uint32_t> basis_alive((size_t)(kCycleBlocks + 1) * BW, 1);
    if ((rc = sv.orthonormalise(sv.W.d(), 0))) return rc;

    const uint32_t m = kCycleBlocks * BW;
    // Thick restart (large n): the 64 best Ritz vectors are kept as blocks 0 and 1 and the last Krylov
    // block (in which all their residuals lie: T Y = Y Theta + V_6 S) continues as block 2, so a cycle
    // extends the space by three blocks for four products, and the kept part of the projection is known:
    // diag(Theta); its coupling to block 2 comes out of the Gram-Schmidt coefficients like any other
    // column. The cycle count stays the same (N = 8000: 21 vs 22) with a third fewer matrix passes, but
    // the projected eigenproblem needs 64 vectors instead of 32: it pays where the passes dominate
    // (N = 32000: 0.54 -> 0.41 s, 16000: 149 -> 141 ms; 8000: 75 -> 78 ms, hence the threshold). Smaller
    // problems restart from the 32 best Ritz vectors alone (six products per cycle).
    uint32_t keep = n >= 12000u ? 2u : 1u;
    if (const char *e = std::getenv("SECEDO_SPECTRAL_KEEP")) keep = std::atoi(e) == 2 && n >= 7 * BW ? 2u : 1u;
    const uint32_t ucols = keep * BW;
    uint32_t kept = 0;  // Ritz blocks at the front of the current basis whose products are not recomputed
    std::vector<double> theta_kept(ucols, 0.0);
    std::vector<double> H((size_t)m * m), theta, U, g;
    secedo_spectral_info inf;
    std::memset(&inf, 0, sizeof(inf));
    std::vector<uint32_t> top(BW);
    std::vector<double> res(BW, 0.0);
    double rr_seconds = 0.0;
    const auto t_solve = std::chrono::steady_clock::now();
    for (uint32_t cycle = 0; cycle < max_cycles; ++cycle) {
        std::fill(H.begin(), H.end(), 0.0);
        for (uint32_t r = 0; r < kept * BW; ++r) H[(size_t)r * m + r] = theta_kept[r];
        for (uint32_t j = kept; j < kCycleBlocks; ++j) {
            if ((rc = sv.product(sv.block(j), sv.W.d()))) return rc;
            ++inf.block_products;
            for (int pass = 0; pass < 2; ++pass) {  // classical Gram-Schmidt, twice
                // the coefficients stay on the device; the host needs them only for the projection H
                double *coef = sv.Gall.d() + (size_t)(j * 2 + pass) * kCycleBlocks * BW * BW;
                SP_TRY(gram(n, sv.Q.d(), sv.blk_stride, j + 1, sv.W.d(), sv.Gp.d(), coef, stream));
                SP_TRY(block_combine(n, sv.Q.d(), sv.blk_stride, j + 1, coef, -1.0, 1.0, sv.W.d(), stream));
            }
            if ((rc = sv.orthonormalise(sv.W.d(), j + 1))) return rc;
        }
        // The one read-back of the cycle: H[blk, j] = sum of the two passes' coefficients of step j, who
        // survived the orthonormalisations, and the R of the last step (for the residuals).
        g.resize((size_t)kCycleBlocks * 2 * kCycleBlocks * BW * BW);
        SP_TRY(hipMemcpyAsync(g.data(), sv.Gall.d(), g.size() * 8, hipMemcpyDeviceToHost, stream));
        SP_TRY(hipMemcpyAsync(basis_alive.data(), sv.alive_dev.p, basis_alive.size() * 4, hipMemcpyDeviceToHost, stream));
        SP_TRY(hipMemcpyAsync(R_last.data(), static_cast<const double *>(sv.Rblk.p) + (size_t)kCycleBlocks * BW * BW,
                              R_last.size() * 8, hipMemcpyDeviceToHost, stream));
        SP_TRY(hipStreamSynchronize(stream));
        for (uint32_t j = kept; j < kCycleBlocks; ++j)
            for (int pass = 0; pass < 2; ++pass)
                for (uint32_t blk = 0; blk <= j; ++blk)
                    for (uint32_t a = 0; a < BW; ++a)
                        for (uint32_t c = 0; c < BW; ++c)
                            H[(size_t)(blk * BW + a) * m + j * BW + c]
                                    += g[(((size_t)(j * 2 + pass) * kCycleBlocks + blk) * BW + a) * BW + c];
        // the projection is symmetric: take the computed block upper triangle, mirror it
        for (uint32_t r = 0; r < m; ++r)
            for (uint32_t c = 0; c < m; ++c)
                if (r / BW > c / BW) H[(size_t)r * m + c] = H[(size_t)c * m + r];
        // a dropped basis column is a zero vector: keep its (zero) Ritz value below the spectrum of T
        for (uint32_t r = 0; r < m; ++r)
            if (!basis_alive[r]) H[(size_t)r * m + r] = -1.0;
        const auto t_rr = std::chrono::steady_clock::now();
        if (!secedo::sym_eig_top((int)m, H, (int)ucols, theta, U, true))  // U: m x ucols, column k = k-th largest; only those values
            return secedo::api_fail(SECEDO_E_LIMIT, "the projected eigenproblem did not converge");
        rr_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rr).count();
        // the largest tau first; residual of a Ritz pair = || R_last u_last || (T V_j = sum_blk V_blk H_blk,j
        // + V_{j+1} R_j, so T y - theta y = V_6 R_5 u_last for y = V u)
        for (uint32_t k = 0; k < BW; ++k) top[k] = m - 1 - k;
        for (uint32_t k = 0; k < BW; ++k) {
            double r2 = 0.0;
            for (uint32_t a = 0; a < BW; ++a) {
                double v = 0.0;
                for (uint32_t c = 0; c < BW; ++c)
                    v += R_last[(size_t)a * BW + c] * U[(size_t)((kCycleBlocks - 1) * BW + c) * ucols + k];
                r2 += v * v;
            }
            res[k] = 2.0 * std::sqrt(r2);  // in units of L = 2 (I - T)
        }
        inf.cycles = cycle + 1;
        inf.max_residual_vectors = 0.0;
        inf.max_residual_values = 0.0;
        for (uint32_t k = 0; k < n_values; ++k) {
            if (k < n_vectors) inf.max_residual_vectors = std::max(inf.max_residual_vectors, res[k]);
            inf.max_residual_values = std::max(inf.max_residual_values, res[k]);
        }
        inf.converged = (inf.max_residual_vectors <= tol && inf.max_residual_values <= tol_values) ? 1u : 0u;
        if (std::getenv("SECEDO_SPECTRAL_TRACE")) {  // diagnostics: residuals of the wanted pairs per cycle
            std::fprintf(stderr, "[spectral] cycle %u:", cycle);
            for (uint32_t k = 0; k < n_values; ++k) std::fprintf(stderr, " %.1e", res[k]);
            std::fprintf(stderr, " | %.1f ms so far, %.1f ms of it in the projected eigenproblem\n",
                         1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_solve).count(),
                         1e3 * rr_seconds);
        }
        // Ritz vectors of the best pairs: Y_q = V U[:, q-th 32 columns] (before any block is overwritten)
        std::vector<double> coeff((size_t)kCycleBlocks * BW * BW);
        const bool last = inf.converged || cycle + 1 == max_cycles;
        for (uint32_t q = 0; q < (last ? 1u : keep); ++q) {
            for (uint32_t blk = 0; blk < kCycleBlocks; ++blk)
                for (uint32_t a = 0; a < BW; ++a)
                    for (uint32_t k = 0; k < BW; ++k)
                        coeff[((size_t)blk * BW + a) * BW + k] = U[(size_t)(blk * BW + a) * ucols + q * BW + k];
            if ((rc = sv.upload_small(coeff))) return rc;
            SP_TRY(block_combine(n, sv.Q.d(), sv.blk_stride, kCycleBlocks, sv.M.d(), 1.0, 0.0,
                                 q == 0 ? sv.W.d() : sv.W2.d(), stream));
        }
        if (last) {
            for (uint32_t k = 0; k < n_values; ++k) eigenvalues[k] = 2.0 * (1.0 - theta[top[k]]);
            SP_TRY(write_vectors(n, sv.W.d(), n_vectors, d_eigenvectors, stream));
            SP_TRY(hipStreamSynchronize(stream));
            break;
        }
        if ((rc = sv.orthonormalise(sv.W.d(), 0))) return rc;
        kept = 0;
        if (keep == 2u) {
            // second Ritz block: orthogonal to the first up to rounding; cleaned like every other block
            for (int pass = 0; pass < 2; ++pass) {
                SP_TRY(gram(n, sv.Q.d(), sv.blk_stride, 1, sv.W2.d(), sv.Gp.d(), sv.Gall.d(), stream));
                SP_TRY(block_combine(n, sv.Q.d(), sv.blk_stride, 1, sv.Gall.d(), -1.0, 1.0, sv.W2.d(), stream));
            }
            if ((rc = sv.orthonormalise(sv.W2.d(), 1))) return rc;
            // the last Krylov block (and who is alive in it) continues as block 2
            SP_TRY(hipMemcpyAsync(sv.block(2), sv.block(kCycleBlocks), sv.blk_stride * 8, hipMemcpyDeviceToDevice, stream));
            SP_TRY(hipMemcpyAsync(static_cast<uint32_t *>(sv.alive_dev.p) + 2 * BW,
                                  static_cast<const uint32_t *>(sv.alive_dev.p) + (size_t)kCycleBlocks * BW, BW * 4,
                                  hipMemcpyDeviceToDevice, stream));
            kept = 2;
            for (uint32_t r = 0; r < ucols; ++r) theta_kept[r] = theta[m - 1 - r];
        }
    }
    if (info) *info = inf;
    return SECEDO_OK;
}

}  // namespace

extern "C" {

int secedo_laplacian_device(const double *d_similarity, uint32_t n, double *d_out, void *stream) {
    if (!d_similarity || !d_out) return secedo::api_fail(SECEDO_E_INVALID_ARG, "null argument");
    if (n == 0) return SECEDO_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Buf s, root, sums;
    SP_TRY(s.alloc((size_t)n * 8));
    SP_TRY(root.alloc((size_t)n * 8));
    SP_TRY(sums.alloc((size_t)n * 8));
    SP_TRY(secedo::spectral::row_sums(d_similarity, n, 0, n, sums.d(), st));
    SP_TRY(secedo::spectral::scale_from_sums(n, sums.d(), s.d(), root.d(), st));
    SP_TRY(secedo::spectral::laplacian(d_similarity, s.d(), n, d_out, st));
    SP_TRY(hipStreamSynchronize(st));  // the scratch is freed on return
    return SECEDO_OK;
}

int secedo_spectral_eigs_device(int device_id, const double *d_similarity, uint32_t n, uint32_t n_values,
                                uint32_t n_vectors, double tol, uint32_t max_cycles, double *eigenvalues,
                                double *d_eigenvectors, secedo_spectral_info *info, void *stream) {
    return solve(device_id, d_similarity, 0, n, n, n_values, n_vectors, tol, max_cycles, eigenvalues, d_eigenvectors,
                 info, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int secedo_spectral_eigs_rows_device(int device_id, const double *d_rows, uint32_t row_begin, uint32_t n_rows,
                                     uint32_t n, uint32_t n_values, uint32_t n_vectors, double tol,
                                     uint32_t max_cycles, double *eigenvalues, double *d_eigenvectors,
                                     secedo_spectral_info *info, secedo_allreduce_sum_fn allreduce,
                                     void *allreduce_ctx, void *stream) {
    return solve(device_id, d_rows, row_begin, n_rows, n, n_values, n_vectors, tol, max_cycles, eigenvalues,
                 d_eigenvectors, info, allreduce, allreduce_ctx, static_cast<hipStream_t>(stream));
}

int secedo_spectral_eigs(int device_id, const double *similarity, uint32_t n, uint32_t n_values,
                         uint32_t n_vectors, double tol, uint32_t max_cycles, double *eigenvalues,
                         double *eigenvectors, secedo_spectral_info *info) {
    if (!similarity) return secedo::api_fail(SECEDO_E_INVALID_ARG, "similarity is null");
    if (n_vectors && !eigenvectors) return secedo::api_fail(SECEDO_E_INVALID_ARG, "eigenvectors is null");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return secedo::api_fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the spectral step has no CPU fallback");
    if (device_id < 0 || device_id >= n_dev) return secedo::api_fail(SECEDO_E_NO_DEVICE, "device id out of range");
    SP_TRY(hipSetDevice(device_id));
    Buf a, v;
    SP_TRY(a.alloc((size_t)n * n * 8));
    SP_TRY(v.alloc((size_t)n * std::max<uint32_t>(n_vectors, 1) * 8));
    SP_TRY(hipMemcpy(a.p, similarity, (size_t)n * n * 8, hipMemcpyHostToDevice));
    const int rc = solve(device_id, a.d(), 0, n, n, n_values, n_vectors, tol, max_cycles, eigenvalues, v.d(), info,
                         nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (n_vectors) SP_TRY(hipMemcpy(eigenvectors, v.p, (size_t)n * n_vectors * 8, hipMemcpyDeviceToHost));
    return SECEDO_OK;
}

}  // extern "C"
