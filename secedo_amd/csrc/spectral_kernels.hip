// spectral_kernels.hip -- gfx950 kernels of the spectral step.
//
//   row_sums         row sums of the similarity matrix -> D^-1/2           (HBM-bound, N^2 read)
//   laplacian        the dense normalised Laplacian, for callers that want the reference's
//                    laplacian() itself (spectral_clustering.cpp:33-52)
//   product_*        Y = (X + D^-1/2 A D^-1/2 X) / 2 for a block of 32 vectors: the matrix is read
//                    ONCE for 32 vectors. This is a tall-skinny fp64 GEMM, so it runs on the matrix
//                    cores (v_mfma_f64_16x16x4_f64); it is bound by the N^2 * 8 bytes of A.
//   gram, block_combine, write_vectors   N x 32 bookkeeping of the block Krylov iteration
//
// The product uses the symmetry of A: Y^T = Z^T A, so the B operand of the MFMA (4 k x 16 columns)
// is 4 rows x 128 contiguous bytes of the row-major matrix -- coalesced without a transpose.
#include "spectral_kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

namespace secedo {
namespace spectral {

namespace {

constexpr int BW = (int)kBlockWidth;
typedef double double4_t __attribute__((ext_vector_type(4)));

// sums[row_begin + r] = sum_j A_rows[r][j] for the local rows (one wave per row)
__global__ __launch_bounds__(256) void k_row_sums(const double *A_rows, uint32_t n, uint32_t row_begin,
                                                 uint32_t n_rows, double *sums) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    const double *a = A_rows + (size_t)row * n;
    double sum = 0.0;
    for (uint32_t j = lane; j < n; j += 64u) sum += a[j];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) sums[row_begin + row] = sum;
}

__global__ __launch_bounds__(256) void k_scale_from_sums(uint32_t n, const double *sums, double *s, double *root) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double sum = sums[i];
    s[i] = sum == 0.0 ? 0.0 : 1.0 / sqrt(sum);  // spectral_clustering.cpp:40-41
    root[i] = sum > 0.0 ? sqrt(sum) : 0.0;
}

__global__ __launch_bounds__(256) void k_laplacian(const double *A, const double *s, uint32_t n, double *out) {
    const size_t total = (size_t)n * n;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t r = (uint32_t)(idx / n), c = (uint32_t)(idx % n);
        out[idx] = (r == c ? 1.0 : 0.0) - s[r] * s[c] * A[idx];
    }
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_init_block(uint32_t n, const double *root, double *X) {
    const size_t total = (size_t)n * BW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t j = (uint32_t)(idx / BW), c = (uint32_t)(idx % BW);
        // 53 random bits -> (-1, 1)
        const double u = (double)(mix64(idx) >> 11) * (1.0 / 9007199254740992.0);
        X[idx] = c == 0 ? root[j] : 2.0 * u - 1.0;
    }
}

// Z[j][c] = s[j] * X[j][c]; rows from n on (padding + 64 rows of slack) are zero
__global__ __launch_bounds__(256) void k_scale_rows(uint32_t n, uint32_t n_pad, const double *s, const double *X,
                                                   double *Z) {
    const size_t total = (size_t)n_pad * BW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t j = (uint32_t)(idx / BW);
        Z[idx] = j < n ? s[j] * X[idx] : 0.0;
    }
}

// P[seg][i][c] = sum over the segment's rows j of Z[j][c] * A[j][i], for the rows this rank holds
// (A_rows = rows [row_begin, row_begin + n_rows) of the matrix).
// A workgroup of four waves covers 128 columns i; a wave owns 32 of them as two interleaved MFMA
// tiles (even and odd columns), so one 16-byte load of A per lane and step of 4 rows feeds four MFMAs
// (2 column tiles x the two 16-wide halves of c). The rows of Z a step needs are the same for the four
// waves: they are staged in LDS 64 rows at a time, double buffered (one barrier per 64 rows).
// (Holding the next 64 rows' matrix values in registers as well costs occupancy and measured slower.)
constexpr uint32_t kProdRows = 64;  // rows of Z per LDS stage; segments are multiples of it
__global__ __launch_bounds__(256) void k_product_partial(const double *A_rows, uint32_t n, uint32_t row_begin,
                                                        uint32_t n_rows, const double *Z, uint32_t seg_rows,
                                                        uint32_t n_pad16, double *P) {
    __shared__ __attribute__((aligned(16))) double Zs[2][kProdRows][BW];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i0 = (blockIdx.x * 4u + wave) * 32u;
    const uint32_t col = lane & 15u, kq = lane >> 4;
    const uint32_t ie = i0 + 2u * col;  // the even column of this lane; ie + 1 the odd one
    const uint32_t ice = min(ie, n - 1u), ico = min(ie + 1u, n - 1u);  // columns past n: computed, never read
    const uint32_t rows_pad = (n_rows + kProdRows - 1u) / kProdRows * kProdRows;
    const uint32_t j_begin = blockIdx.y * seg_rows, j_end = min(rows_pad, j_begin + seg_rows);
    double4_t acc_e0 = {0.0, 0.0, 0.0, 0.0}, acc_e1 = acc_e0, acc_o0 = acc_e0, acc_o1 = acc_e0;
    // stage: 64 rows x 32 doubles = 16 KiB, 64 bytes per thread (Z has kProdRows zero rows of slack)
    auto stage = [&](uint32_t buf, uint32_t j0) {
        const double2 *src = reinterpret_cast<const double2 *>(Z + (size_t)(row_begin + j0) * BW) + tid * 4u;
        double2 *dst = reinterpret_cast<double2 *>(&Zs[buf][0][0]) + tid * 4u;
#pragma unroll
        for (int u = 0; u < 4; ++u) dst[u] = src[u];
    };
    // 64 rows against the staged Z. FAST: the wave's 32 columns and the 64 rows all lie inside the block
    // and the row pitch keeps 16-byte alignment -- straight-line code, the four loads of a group of 16
    // rows in flight together. Otherwise every access is clamped and masked (edges only).
    auto rows64 = [&](uint32_t j0, uint32_t buf, auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll 2
        for (uint32_t q = 0; q < kProdRows; q += 16u) {
            double ae[4], ao[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t jl = j0 + q + 4u * u + kq;
                if (FAST) {
                    const double2 v = *reinterpret_cast<const double2 *>(A_rows + (size_t)jl * n + ie);
                    ae[u] = v.x;
                    ao[u] = v.y;
                } else {
                    const double *row = A_rows + (size_t)min(jl, n_rows - 1u) * n;
                    const double ve = row[ice], vo = row[ico];
                    ae[u] = jl < n_rows ? ve : 0.0;  // local rows past the block contribute nothing
                    ao[u] = jl < n_rows ? vo : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double z0 = Zs[buf][q + 4u * u + kq][col], z1 = Zs[buf][q + 4u * u + kq][16u + col];
                acc_e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, ae[u], acc_e0, 0, 0, 0);
                acc_e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, ae[u], acc_e1, 0, 0, 0);
                acc_o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, ao[u], acc_o0, 0, 0, 0);
                acc_o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, ao[u], acc_o1, 0, 0, 0);
            }
        }
    };
    const bool cols_inside = i0 + 32u <= n && (n & 1u) == 0u;  // wave-uniform
    if (j_begin < j_end) stage(0, j_begin);
    __syncthreads();
    uint32_t buf = 0;
    for (uint32_t j0 = j_begin; j0 < j_end; j0 += kProdRows, buf ^= 1u) {
        if (j0 + kProdRows < j_end) stage(buf ^ 1u, j0 + kProdRows);
        if (i0 < n_pad16) {
            if (cols_inside && j0 + kProdRows <= n_rows) rows64(j0, buf, std::true_type{});
            else rows64(j0, buf, std::false_type{});
        }
        __syncthreads();  // the next stage is complete, this one free to be overwritten
    }
    if (i0 >= n_pad16) return;
    // C/D layout of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * reg
    if (ie < n_pad16) {
        double *p = P + ((size_t)blockIdx.y * n_pad16 + ie) * BW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[kq + 4u * r] = acc_e0[r];
            p[16u + kq + 4u * r] = acc_e1[r];
        }
    }
    if (ie + 1u < n_pad16) {
        double *p = P + ((size_t)blockIdx.y * n_pad16 + ie + 1u) * BW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[kq + 4u * r] = acc_o0[r];
            p[16u + kq + 4u * r] = acc_o1[r];
        }
    }
}

// Ypart[i][c] = sum_seg P[seg][i][c], segments added in order; or, when there is nothing to add from other
// ranks (Y != null), the finished product at once: Y[i][c] = (X[i][c] + s[i] * sum) / 2 (k_product_finish)
__global__ __launch_bounds__(256) void k_product_reduce(uint32_t n, uint32_t n_seg, uint32_t n_pad16, const double *P,
                                                       double *Ypart, const double *s, const double *X, double *Y) {
    const size_t total = (size_t)n * BW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        double sum = 0.0;
        for (uint32_t g = 0; g < n_seg; ++g) sum += P[(size_t)g * n_pad16 * BW + idx];
        if (Y) Y[idx] = 0.5 * (X[idx] + s[idx / BW] * sum);
        else Ypart[idx] = sum;
    }
}

// Y[i][c] = (X[i][c] + s[i] * Ysum[i][c]) / 2
__global__ __launch_bounds__(256) void k_product_finish(uint32_t n, const double *Ysum, const double *s,
                                                       const double *X, double *Y) {
    const size_t total = (size_t)n * BW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
        Y[idx] = 0.5 * (X[idx] + s[idx / BW] * Ysum[idx]);
}

// Gp[chunk][blk][a][c] = sum over the chunk's rows j of Q[blk][j][a] * W[j][c]
__global__ __launch_bounds__(256) void k_gram_partial(uint32_t n, const double *Q, size_t blk_stride,
                                                     const double *W, uint32_t nblk, double *Gp) {
    __shared__ double Qs[kGramChunk][BW];
    __shared__ double Ws[kGramChunk][BW];
    const uint32_t chunk = blockIdx.x, blk = blockIdx.y;
    const uint32_t j0 = chunk * kGramChunk, rows = min(kGramChunk, n - j0);
    const double *q = Q + blk * blk_stride + (size_t)j0 * BW;
    const double *w = W + (size_t)j0 * BW;
    for (uint32_t i = threadIdx.x; i < rows * BW; i += 256u) {
        (&Qs[0][0])[i] = q[i];
        (&Ws[0][0])[i] = w[i];
    }
    __syncthreads();
    const uint32_t a = threadIdx.x >> 3, c0 = (threadIdx.x & 7u) * 4u;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (uint32_t j = 0; j < rows; ++j) {
        const double qa = Qs[j][a];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += qa * Ws[j][c0 + u];
    }
    double *out = Gp + ((size_t)chunk * nblk + blk) * BW * BW + a * BW + c0;
#pragma unroll
    for (int u = 0; u < 4; ++u) out[u] = acc[u];
}

// G[blk] = sum over chunks of Gp[chunk][blk], chunks added in order (deterministic). One thread per
// element, four workgroups per block; the loads of eight chunks are in flight together.
__global__ __launch_bounds__(256) void k_gram_reduce(uint32_t n_chunks, uint32_t nblk, const double *Gp, double *G) {
    const uint32_t blk = blockIdx.x >> 2, e = (blockIdx.x & 3u) * 256u + threadIdx.x;
    const size_t stride = (size_t)nblk * BW * BW;
    const double *p = Gp + (size_t)blk * BW * BW + e;
    double sum = 0.0;
    uint32_t ch = 0;
    for (; ch + 8 <= n_chunks; ch += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(ch + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; ch < n_chunks; ++ch) sum += p[(size_t)ch * stride];
    G[(size_t)blk * BW * BW + e] = sum;
}

// out[j][c] = beta * out[j][c] + alpha * sum_blk sum_a Q[blk][j][a] * M[blk][a][c]
// 16 rows per workgroup (enough workgroups to cover the chip at N = 4096), 2 columns per thread
__global__ __launch_bounds__(256) void k_block_combine(uint32_t n, const double *Q, size_t blk_stride, uint32_t nblk,
                                                      const double *M, double alpha, double beta, double *out) {
    constexpr uint32_t ROWS = 16;
    __shared__ double Ms[BW][BW];
    __shared__ double Qs[ROWS][BW + 1];
    const uint32_t j0 = blockIdx.x * ROWS, rows = min(ROWS, n - j0);
    const uint32_t jr = threadIdx.x >> 4, c0 = (threadIdx.x & 15u) * 2u;
    double acc0 = 0.0, acc1 = 0.0;
    for (uint32_t blk = 0; blk < nblk; ++blk) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < BW * BW; i += 256u) (&Ms[0][0])[i] = M[(size_t)blk * BW * BW + i];
        const double *q = Q + blk * blk_stride + (size_t)j0 * BW;
        for (uint32_t i = threadIdx.x; i < rows * BW; i += 256u) Qs[i / BW][i % BW] = q[i];
        __syncthreads();
        if (jr < rows) {
#pragma unroll 8
            for (uint32_t a = 0; a < (uint32_t)BW; ++a) {
                const double qa = Qs[jr][a];
                acc0 += qa * Ms[a][c0];
                acc1 += qa * Ms[a][c0 + 1];
            }
        }
    }
    if (jr < rows) {
        double *o = out + (size_t)(j0 + jr) * BW + c0;
        o[0] = (beta == 0.0 ? 0.0 : beta * o[0]) + alpha * acc0;
        o[1] = (beta == 0.0 ? 0.0 : beta * o[1]) + alpha * acc1;
    }
}

// one workgroup per output column
__global__ __launch_bounds__(256) void k_write_vectors(uint32_t n, const double *Y, double *out) {
    __shared__ double s_norm[256], s_best[256];
    __shared__ uint32_t s_idx[256];
    const uint32_t c = blockIdx.x;
    double norm2 = 0.0, best = -1.0;
    uint32_t best_i = 0;
    for (uint32_t j = threadIdx.x; j < n; j += 256u) {
        const double v = Y[(size_t)j * BW + c];
        norm2 += v * v;
        if (fabs(v) > best) {  // strided ascending: the first hit of a thread is its lowest index
            best = fabs(v);
            best_i = j;
        }
    }
    s_norm[threadIdx.x] = norm2;
    s_best[threadIdx.x] = best;
    s_idx[threadIdx.x] = best_i;
    __syncthreads();
    for (uint32_t half = 128; half > 0; half >>= 1) {
        if (threadIdx.x < half) {
            s_norm[threadIdx.x] += s_norm[threadIdx.x + half];
            const double ob = s_best[threadIdx.x + half];
            const uint32_t oi = s_idx[threadIdx.x + half];
            if (ob > s_best[threadIdx.x] || (ob == s_best[threadIdx.x] && oi < s_idx[threadIdx.x])) {
                s_best[threadIdx.x] = ob;
                s_idx[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const double norm = sqrt(s_norm[0]);
    double scale = norm > 0.0 ? 1.0 / norm : 0.0;
    if (Y[(size_t)s_idx[0] * BW + c] < 0.0) scale = -scale;
    for (uint32_t j = threadIdx.x; j < n; j += 256u) out[(size_t)c * n + j] = scale * Y[(size_t)j * BW + c];
}

inline uint32_t grid_for(size_t total) {
    return (uint32_t)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 256 * 16));
}

// Cholesky factor of a 32 x 32 Gram matrix with column dropping, its inverse, and the product with the
// factor of an earlier pass -- one workgroup, one thread per matrix element (on the host this cost two
// synchronisations per block step). A column whose norm is below 1e-13, or whose part outside the span
// of the columns before it is below 1e-5 of its norm, is dropped: it keeps its coefficients on the
// surviving directions in R, its row of R and its row and column of R^-1 are zero, so the
// orthonormalised block W R^-1 has a zero column there.
// Right-looking elimination on the matrix scaled to unit diagonal. Thread (r, c) keeps element (r, c) in
// a register; only the pivot row goes through LDS, one barrier per pivot; who is alive is recomputed by
// every thread from the same values. The inverse is built in registers too, a row per thread.
__global__ __launch_bounds__(1024) void k_cholesky_drop(const double *G, const double *R_prev, double *Rinv_out,
                                                       double *R_out, uint32_t *alive_out) {
    constexpr int N = 32;
    __shared__ double prow[N][N];  // prow[k][c]: row k of the scaled matrix when k becomes the pivot
    __shared__ double R[N][N + 1];
    __shared__ double rdiag_inv[N];
    const int tid = threadIdx.x, r = tid / N, c = tid % N;
    const double g_rr = G[r * N + r], g_cc = G[c * N + c];
    // lanes 0..31 of every wave hold c = 0..31: norm above 1e-13, else an exhausted direction
    uint32_t alive = (uint32_t)__ballot(g_cc > 1e-26);
    const double d_r = g_rr > 1e-26 ? sqrt(g_rr) : 0.0, d_c = g_cc > 1e-26 ? sqrt(g_cc) : 0.0;
    double x = 0.0;  // element (r, c) of the scaled matrix, upper triangle
    if (r == c) x = 1.0;
    else if (c > r && ((alive >> r) & 1u) && ((alive >> c) & 1u)) x = G[tid] / (d_r * d_c);
    double s_own = 0.0;  // element (r, c) of the factor of the scaled matrix
    if (r == 0) prow[0][c] = x;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        // every thread takes the same decision from the same LDS word
        const double piv = prow[k][k];
        const bool use = ((alive >> k) & 1u) && piv > 1e-10;  // else exhausted, or dependent on the columns before
        if (((alive >> k) & 1u) && !use) alive &= ~(1u << k);
        if (use) {
            // one reciprocal square root instead of a square root and two divisions on the critical path
            // of 32 dependent pivots (a last-bit difference in R is immaterial: the QR is done twice)
            const double inv_sk = rsqrt(piv);
            const double skr = r > k && ((alive >> r) & 1u) ? prow[k][r] * inv_sk : 0.0;
            const double skc = c > k && ((alive >> c) & 1u) ? prow[k][c] * inv_sk : 0.0;
            if (r == k) s_own = (c == k) ? piv * inv_sk : skc;
            if (r > k && c >= r) x -= skr * skc;
        }
        if (r == k + 1) prow[k + 1][c] = x;  // the next pivot row is final now
        __syncthreads();
    }
    R[r][c] = (c >= r && ((alive >> r) & 1u)) ? s_own * d_c : 0.0;
    __syncthreads();
    if (tid < N) rdiag_inv[tid] = ((alive >> tid) & 1u) ? 1.0 / R[tid][tid] : 0.0;
    __syncthreads();
    // R^-1 on the surviving triangle from X R = I: thread `row` (32 threads) builds row `row` of X left to
    // right, X[row][c] = (delta - sum_{t < c} X[row][t] R[t][c]) / R[c][c], in registers (the loops are
    // unrolled; every thread reads the same R[t][c]: an LDS broadcast) -- no cross-lane sums
    if (tid < N) {
        const int row = tid;
        const bool row_alive = (alive >> row) & 1u;
        double xr[N];
#pragma unroll
        for (int col = 0; col < N; ++col) {
            double v = col == row ? 1.0 : 0.0;
#pragma unroll
            for (int t = 0; t < col; ++t) v -= xr[t] * R[t][col];
            xr[col] = (col >= row && row_alive && ((alive >> col) & 1u)) ? v * rdiag_inv[col] : 0.0;
        }
#pragma unroll
        for (int col = 0; col < N; ++col) Rinv_out[row * N + col] = xr[col];
    }
    double v = R[r][c];
    if (R_prev) {  // R of the two passes together: this pass's factor times the first one's
        v = 0.0;
        for (int t = 0; t < N; ++t) v += R[r][t] * R_prev[t * N + c];
    }
    R_out[tid] = v;
    if (tid < N && alive_out) alive_out[tid] = (alive >> tid) & 1u;
}

}  // namespace

hipError_t row_sums(const double *A_rows, uint32_t n, uint32_t row_begin, uint32_t n_rows, double *sums,
                    hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_row_sums, dim3((n_rows + 3) / 4), dim3(256), 0, stream, A_rows, n, row_begin, n_rows, sums);
    return hipGetLastError();
}

hipError_t scale_from_sums(uint32_t n, const double *sums, double *s, double *root, hipStream_t stream) {
    hipLaunchKernelGGL(k_scale_from_sums, dim3((n + 255) / 256), dim3(256), 0, stream, n, sums, s, root);
    return hipGetLastError();
}

hipError_t laplacian(const double *A, const double *s, uint32_t n, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(k_laplacian, dim3(grid_for((size_t)n * n)), dim3(256), 0, stream, A, s, n, out);
    return hipGetLastError();
}

hipError_t init_block(uint32_t n, const double *root, double *X, hipStream_t stream) {
    hipLaunchKernelGGL(k_init_block, dim3(grid_for((size_t)n * BW)), dim3(256), 0, stream, n, root, X);
    return hipGetLastError();
}

uint32_t product_segments(uint32_t n, uint32_t n_rows) {
    // enough waves to fill 256 CUs: a workgroup column covers 64 matrix columns
    const uint32_t cols = (pad16(n) + 127u) / 128u;
    const uint32_t want = (2048u + cols - 1u) / cols;
    const uint32_t most = std::max(1u, (n_rows + 127u) / 128u);  // at least 128 rows per segment
    return std::max(1u, std::min({want, 64u, most}));
}

hipError_t product_partial(const double *A_rows, uint32_t n, uint32_t row_begin, uint32_t n_rows, const double *s,
                           const double *X, double *Z, double *P, double *Ypart, double *Y_finished,
                           hipStream_t stream) {
    const uint32_t nz = pad16(n) + kProdRows, n16 = pad16(n), n_seg = product_segments(n, n_rows);
    const uint32_t seg_rows = ((n_rows + n_seg - 1u) / n_seg + kProdRows - 1u) / kProdRows * kProdRows;
    hipLaunchKernelGGL(k_scale_rows, dim3(grid_for((size_t)nz * BW)), dim3(256), 0, stream, n, nz, s, X, Z);
    if (n_rows)
        hipLaunchKernelGGL(k_product_partial, dim3((n16 + 127u) / 128u, n_seg), dim3(256), 0, stream, A_rows, n, row_begin,
                           n_rows, Z, seg_rows, n16, P);
    hipLaunchKernelGGL(k_product_reduce, dim3(grid_for((size_t)n * BW)), dim3(256), 0, stream, n, n_rows ? n_seg : 0u,
                       n16, P, Ypart, s, X, Y_finished);
    return hipGetLastError();
}

hipError_t product_finish(uint32_t n, const double *s, const double *X, const double *Ysum, double *Y,
                          hipStream_t stream) {
    hipLaunchKernelGGL(k_product_finish, dim3(grid_for((size_t)n * BW)), dim3(256), 0, stream, n, Ysum, s, X, Y);
    return hipGetLastError();
}

uint32_t gram_chunks(uint32_t n) { return (n + kGramChunk - 1u) / kGramChunk; }

hipError_t gram(uint32_t n, const double *Q, size_t blk_stride, uint32_t nblk, const double *W, double *Gp,
                double *G, hipStream_t stream) {
    const uint32_t chunks = gram_chunks(n);
    hipLaunchKernelGGL(k_gram_partial, dim3(chunks, nblk), dim3(256), 0, stream, n, Q, blk_stride, W, nblk, Gp);
    hipLaunchKernelGGL(k_gram_reduce, dim3(nblk * 4u), dim3(256), 0, stream, chunks, nblk, Gp, G);
    return hipGetLastError();
}

hipError_t cholesky_drop(const double *G, const double *R_prev, double *Rinv, double *R, uint32_t *alive,
                         hipStream_t stream) {
    hipLaunchKernelGGL(k_cholesky_drop, dim3(1), dim3(1024), 0, stream, G, R_prev, Rinv, R, alive);
    return hipGetLastError();
}

hipError_t block_combine(uint32_t n, const double *Q, size_t blk_stride, uint32_t nblk, const double *M,
                         double alpha, double beta, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(k_block_combine, dim3((n + 15u) / 16u), dim3(256), 0, stream, n, Q, blk_stride, nblk, M, alpha,
                       beta, out);
    return hipGetLastError();
}

hipError_t write_vectors(uint32_t n, const double *Y, uint32_t k, double *out, hipStream_t stream) {
    if (k == 0) return hipSuccess;
    hipLaunchKernelGGL(k_write_vectors, dim3(k), dim3(256), 0, stream, n, Y, out);
    return hipGetLastError();
}

}  // namespace spectral
}  // namespace secedo
