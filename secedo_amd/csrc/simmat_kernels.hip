// simmat_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the similarity-matrix path.
//
//   accumulate_tiles   the quadratic part: compare_with_reads + apply_updates
//                      (reference: similarity_matrix.cpp:189-243, :246-254). One workgroup owns one
//                      B x B cell-pair tile of the output for a range of loci and keeps it as int64
//                      fixed-point accumulators in LDS (ds_add_u64); entry records stream in from
//                      HBM/L2 as 16-byte loads. No MFMA: this is an indexed gather/accumulate.
//   reduce_max         max of D over the matrix, for ADD_MIN / SCALE_MAX_1 (:271-293)
//   write_matrix       tile-major accumulator -> dense row-major fp64 matrix, mirrored, normalised
//
// 64-wide wavefronts throughout; nothing here depends on workgroup dispatch order or XCD placement.
#include "simmat_kernels.hpp"

#include <hip/hip_runtime.h>

namespace secedo {

namespace {

constexpr uint32_t META_TAIL = 1u << 18;
constexpr uint32_t META_PREV_OVF = 1u << 19;
constexpr uint32_t META_NEXT_OVF = 1u << 20;
constexpr int LUT_DIM = 65;

__device__ __forceinline__ double log_add(double a, double b) {
    const double hi = fmax(a, b), lo = fmin(a, b);
    return hi + log1p(exp(lo - hi));
}

// D(x_s, x_d) outside the table: closed form in log space (llr_table.hpp)
__device__ __noinline__ long long llr_fixed_device(const LlrModelDev &m, uint32_t xs, uint32_t xd,
                                                   int scale_log2) {
    const double s = xs, d = xd;
    const double diff = log_add(s * m.ln_u1 + d * m.ln_v1, s * m.ln_u2 + d * m.ln_v2);
    const double same = log_add(s * m.ln_w1 + d * m.ln_z1, s * m.ln_w2 + d * m.ln_z2);
    return llrint(ldexp(diff - same, scale_log2));
}

// Slow path (a window overflowed on the same side for both reads): merge-walk the two reads'
// kept entries (reference: similarity_matrix.cpp:223-229). Returns the first shared locus, or
// 0xFFFFFFFF if none.
__device__ __noinline__ uint32_t merge_walk(const AccumulateArgs &a, uint32_t r1, uint32_t r2,
                                            uint32_t *xs_out, uint32_t *xd_out) {
    uint32_t i1 = a.read_off[r1], e1 = a.read_off[r1 + 1];
    uint32_t i2 = a.read_off[r2], e2 = a.read_off[r2 + 1];
    uint32_t xs = 0, xd = 0, first = 0xFFFFFFFFu;
    while (i1 < e1 && i2 < e2) {
        const uint32_t l1 = a.read_locus[i1], l2 = a.read_locus[i2];
        if (l1 == l2) {
            if (first == 0xFFFFFFFFu) first = l1;
            if (a.read_base[i1] == a.read_base[i2]) ++xs; else ++xd;
            ++i1; ++i2;
        } else if (l1 < l2) {
            ++i1;
        } else {
            ++i2;
        }
    }
    *xs_out = xs;
    *xd_out = xd;
    return first;
}

template <int B, int THREADS>
__global__ __launch_bounds__(THREADS) void accumulate_tiles(const AccumulateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long tile[];  // B*B

    const uint32_t t = a.tile_begin + blockIdx.x / a.n_chunks;
    const uint32_t chunk = blockIdx.x % a.n_chunks;
    const uint32_t I = a.tile_row[t], J = a.tile_col[t];
    const uint32_t tid = threadIdx.x;

    for (uint32_t i = tid; i < B * B; i += THREADS) tile[i] = 0ull;
    __syncthreads();

    const uint32_t l0 = chunk * a.chunk_loci;
    const uint32_t l1 = min(a.num_loci, l0 + a.chunk_loci);
    const size_t rowI = (size_t)I * a.stride, rowJ = (size_t)J * a.stride;
    unsigned long long n_updates = 0, n_pairs = 0;

    if (l0 < l1) {
        const uint32_t i_begin = a.blk_off[rowI + l0], i_end = a.blk_off[rowI + l1];
        const long long d10 = a.lut[1 * LUT_DIM + 0], d01 = a.lut[0 * LUT_DIM + 1];
        for (uint32_t e1 = i_begin + tid; e1 < i_end; e1 += THREADS) {
            const uint4 A1 = a.entry_a[e1];
            const uint32_t l = A1.w;
            uint32_t j0, j1;
            if (I == J) {  // pairs inside one block: every unordered pair once
                j0 = e1 + 1;
                j1 = a.blk_off[rowI + l + 1];
            } else {
                j0 = a.blk_off[rowJ + l];
                j1 = a.blk_off[rowJ + l + 1];
            }
            const uint32_t c1 = A1.x & 0xFFFFu;
            const uint32_t row = (c1 - I * B) * B;
            for (uint32_t e2 = j0; e2 < j1; ++e2) {
                const uint4 A2 = a.entry_a[e2];
                const uint32_t c2 = A2.x & 0xFFFFu;
                if (c1 == c2) continue;                    // same cell (:215)
                if (A1.x & A2.x & META_TAIL) continue;     // neither read was ever flushed (:407-408)
                ++n_updates;
                if (A1.y & A2.y) continue;                 // an earlier shared locus owns this pair
                const bool same = (((A1.x ^ A2.x) >> 16) & 3u) == 0u;
                long long v;
                if ((A1.x & A2.x & (META_PREV_OVF | META_NEXT_OVF)) == 0u) {
                    const uint32_t shared = A1.z & A2.z;
                    if (shared == 0u) {
                        v = same ? d10 : d01;              // the pair shares this locus only
                    } else {                               // joint (x_s, x_d) term from the windows
                        const uint4 B1 = a.entry_b[e1];
                        const uint4 B2 = a.entry_b[e2];
                        const uint32_t diff = ((B1.x ^ B2.x) | (B1.y ^ B2.y)) & shared;
                        const uint32_t xd = __popc(diff) + (same ? 0u : 1u);
                        const uint32_t xs = __popc(shared) - __popc(diff) + (same ? 1u : 0u);
                        v = a.lut[xs * LUT_DIM + xd];
                    }
                } else {
                    uint32_t xs, xd;
                    const uint32_t first = merge_walk(a, a.entry_b[e1].z, a.entry_b[e2].z, &xs, &xd);
                    if (first != l) continue;
                    v = (xs < LUT_DIM && xd < LUT_DIM) ? a.lut[xs * LUT_DIM + xd]
                                                       : llr_fixed_device(a.model, xs, xd, a.scale_log2);
                }
                ++n_pairs;
                atomicAdd(&tile[row + (c2 - J * B)], (unsigned long long)v);
            }
        }
    }
    __syncthreads();

    unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.acc) + (size_t)t * B * B;
    if (a.n_chunks == 1) {
        for (uint32_t i = tid; i < B * B; i += THREADS) dst[i] = tile[i];
    } else {
        for (uint32_t i = tid; i < B * B; i += THREADS) {
            const unsigned long long v = tile[i];
            if (v) atomicAdd(&dst[i], v);
        }
    }

    // work counters: wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_updates += __shfl_down(n_updates, off);
        n_pairs += __shfl_down(n_pairs, off);
    }
    if ((tid & 63u) == 0u && (n_updates | n_pairs)) {
        atomicAdd(&a.counters[0], n_updates);
        atomicAdd(&a.counters[1], n_pairs);
    }
}

// upper-triangular tile index of (I <= J)
__device__ __forceinline__ size_t tile_index(uint32_t I, uint32_t J, uint32_t nb) {
    return (size_t)I * nb - (size_t)I * (I - 1) / 2 + (J - I);
}

// D[i][j] (i != j) from the tile-major accumulator, as an exact integer
template <int B>
__device__ __forceinline__ long long acc_value(const long long *acc, uint32_t nb, uint32_t i, uint32_t j) {
    uint32_t I = i / B, J = j / B, r = i % B, c = j % B;
    if (I > J) {
        uint32_t tI = I; I = J; J = tI;
        uint32_t tr = r; r = c; c = tr;
    }
    const long long *tile = acc + tile_index(I, J, nb) * B * B;
    long long v = tile[r * B + c];
    if (I == J) v += tile[c * B + r];  // a diagonal tile holds each pair in either orientation
    return v;
}

// max over i < j of D[i][j], clamped at 0 (the diagonal is zero): bits of a non-negative double
// order like unsigned integers, so atomicMax on the bit pattern is exact
template <int B>
__global__ __launch_bounds__(256) void reduce_max(const long long *acc, uint32_t n, uint32_t nb,
                                                  double scale, unsigned long long *out_bits) {
    const size_t total = (size_t)n * n;
    double best = 0.0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t i = idx / n, j = idx % n;
        if (i < j) {
            const double d = (double)acc_value<B>(acc, nb, i, j) * scale;
            best = fmax(best, d);
        }
    }
    for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_down(best, off));
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
        atomicMax(out_bits, (unsigned long long)__double_as_longlong(best));
    }
}

// mode: 0 ADD_MIN, 1 EXPONENTIATE, 2 SCALE_MAX_1, 3 raw D (reference: similarity_matrix.cpp:271-293)
template <int B>
__global__ __launch_bounds__(256) void write_matrix(const long long *acc, uint32_t n, uint32_t nb,
                                                    double scale, int mode,
                                                    const unsigned long long *max_bits, double *out) {
    const size_t total = (size_t)n * n;
    const double mx = (mode == 0 || mode == 2) ? __longlong_as_double((long long)*max_bits) : 0.0;
    // ADD_MIN: sim = -D; sim += |min(sim)|, and min(sim) = -max(D) with the zero diagonal included
    // SCALE_MAX_1: sim = D * (1 / max(D)) with the zero diagonal included (1/0 = inf as in the reference)
    const double add = fabs(-mx);
    const double inv = 1.0 / mx;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t i = idx / n, j = idx % n;
        double v = 0.0;
        if (i != j) {
            const double d = (double)acc_value<B>(acc, nb, i, j) * scale;
            switch (mode) {
                case 0: v = (d * -1.0) + add; break;
                case 1: v = 1.0 / (exp(d) + 1.0); break;
                case 2: v = d * inv; break;
                default: v = d; break;
            }
        }
        out[idx] = v;
    }
}

}  // namespace

hipError_t launch_accumulate(const AccumulateArgs &args, uint32_t block_cells, uint32_t n_tiles,
                             hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    const uint32_t grid = n_tiles * args.n_chunks;
    if (block_cells == 128) {
        constexpr int B = 128, T = 1024;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_tiles<B, T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, B * B * 8);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((accumulate_tiles<B, T>), dim3(grid), dim3(T), B * B * 8, stream, args);
    } else {
        constexpr int B = 64, T = 256;
        hipLaunchKernelGGL((accumulate_tiles<B, T>), dim3(grid), dim3(T), B * B * 8, stream, args);
    }
    return hipGetLastError();
}

hipError_t launch_finalize(const int64_t *acc, uint32_t n, uint32_t nb, uint32_t block_cells,
                           int scale_log2, int mode, unsigned long long *d_max_bits, double *out,
                           hipStream_t stream) {
    const double scale = ldexp(1.0, -scale_log2);
    const size_t total = (size_t)n * n;
    const uint32_t grid = (uint32_t)std::min<size_t>((total + 255) / 256, 256 * 8);
    const long long *a = reinterpret_cast<const long long *>(acc);
    if (mode == 0 || mode == 2) {
        hipError_t e = hipMemsetAsync(d_max_bits, 0, sizeof(unsigned long long), stream);
        if (e != hipSuccess) return e;
        if (block_cells == 128) {
            hipLaunchKernelGGL((reduce_max<128>), dim3(grid), dim3(256), 0, stream, a, n, nb, scale, d_max_bits);
        } else {
            hipLaunchKernelGGL((reduce_max<64>), dim3(grid), dim3(256), 0, stream, a, n, nb, scale, d_max_bits);
        }
    }
    if (block_cells == 128) {
        hipLaunchKernelGGL((write_matrix<128>), dim3(grid), dim3(256), 0, stream, a, n, nb, scale, mode, d_max_bits, out);
    } else {
        hipLaunchKernelGGL((write_matrix<64>), dim3(grid), dim3(256), 0, stream, a, n, nb, scale, mode, d_max_bits, out);
    }
    return hipGetLastError();
}

}  // namespace secedo
