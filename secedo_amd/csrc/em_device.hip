// em_device.hip -- EM refinement of a two-way split on the GPU (include/secedo_em.h; reference
// expectation_maximization.cpp:19-161).
//
// Per iteration the reference walks the loci once: two weighted base compositions per locus (the
// "cluster centres", :19-39) and, per entry, two scatter-adds into per-cell sums (:77-80). Here:
//   once      entries regrouped by cell (stable radix sort of (cell, locus << 2 | base)): the
//             scatter-add becomes a gather, no atomics, the summation order is fixed;
//   centres   one wave per locus: 8 weighted counts by wave reduction, the four logs of each centre;
//   cell sums one wave per cell over its (locus, base) list, reading the centres (L x 64 bytes,
//             L2 resident);
//   E step    one workgroup: priors, odds, the new probabilities, "nothing moved by 1e-2" (:100-123).
// HBM-bound: 6 B per entry for the centres + 4 B per entry for the sums per iteration.
#include "secedo_em.h"
#include "secedo_simmat.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace secedo {
int api_fail(int code, const std::string &msg);  // simmat_api.cpp
}

namespace {

#define EM_TRY(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return secedo::api_fail(SECEDO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

struct Buf {
    void *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

// Scratch of one refinement: one allocation, carved up. The reference refines once per sub-cluster of
// its recursion (spectral_clustering.cpp:425-426), and hipMalloc + hipFree of nine buffers cost more than
// the kernels: the allocation is kept per device between calls (grown when too small) and handed to one
// caller at a time; a concurrent caller allocates its own. secedo_simmat_release_cache() frees it,
// SECEDO_ONE_SHOT_CACHE=0 turns the pool off.
struct Arena {
    void *p = nullptr;
    size_t bytes = 0;
    bool busy = false;
    hipError_t ensure(size_t n) {
        if (p && n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        const size_t want = n + n / 8 + 256;
        const hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};
std::mutex g_arena_mutex;
std::map<int, Arena> g_arenas;  // leaked at exit on purpose: the HIP runtime may be gone by then

struct ArenaLease {
    Arena own;
    Arena *a = &own;
    explicit ArenaLease(int device) {
        const char *e = std::getenv("SECEDO_ONE_SHOT_CACHE");
        if (e && std::strcmp(e, "0") == 0) return;
        std::lock_guard<std::mutex> lock(g_arena_mutex);
        Arena &slot = g_arenas[device];
        if (!slot.busy) {
            slot.busy = true;
            a = &slot;
        }
    }
    ~ArenaLease() {
        if (a == &own) {
            own.release();
        } else {
            std::lock_guard<std::mutex> lock(g_arena_mutex);
            a->busy = false;
        }
    }
    ArenaLease(const ArenaLease &) = delete;
    ArenaLease &operator=(const ArenaLease &) = delete;
};

struct Flags {
    uint32_t error;  // 1: group id >= n_cells (prob index), 2: group outside id_to_pos / position >= n_cells
    uint32_t done;
    uint32_t iterations;  // E-steps run (counted on the device: iterations are launched in batches)
};

__device__ __forceinline__ uint32_t id_base_at(const uint16_t *b16, const uint32_t *b32, uint64_t e) {
    return b16 ? (uint32_t)b16[e] : b32[e];
}

// (cell, locus << 2 | base) per entry, one wave per locus; validates the two mappings
__global__ __launch_bounds__(256) void k_em_keys(const uint64_t *off, uint32_t n_loci, const uint16_t *b16,
                                                const uint32_t *b32, const uint32_t *id_to_pos,
                                                uint32_t n_groups, uint32_t n_cells, uint32_t *key,
                                                uint32_t *val, Flags *flags) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (uint32_t l = wave; l < n_loci; l += n_waves) {
        const uint64_t b = off[l], e = off[l + 1];
        for (uint64_t i = b + lane; i < e; i += 64u) {
            const uint32_t ib = id_base_at(b16, b32, i), group = ib >> 2;
            uint32_t cell = 0;
            if (group >= n_cells) flags->error = 1;
            if (group >= n_groups || id_to_pos[group] >= n_cells) flags->error = flags->error ? flags->error : 2;
            else cell = id_to_pos[group];
            key[i] = cell;
            val[i] = (l << 2) | (ib & 3u);
        }
    }
}

__global__ void k_em_cell_offsets(const uint32_t *sorted_key, uint32_t n, uint32_t n_cells, uint32_t *cell_off) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c <= n_cells; c += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = n;  // first position with key >= c
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (sorted_key[mid] < c) lo = mid + 1; else hi = mid;
        }
        cell_off[c] = lo;
    }
}

// cluster_center (:19-39) for both clusters: centres[l][0..3] = log centre A, [4..7] = log centre B
__global__ __launch_bounds__(256) void k_em_centres(const uint64_t *off, uint32_t n_loci, const uint16_t *b16,
                                                   const uint32_t *b32, const double *prob_b, double theta,
                                                   double *centres, const Flags *flags) {
    if (flags->done) return;  // settled earlier in this batch of iterations
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (uint32_t l = wave; l < n_loci; l += n_waves) {
        const uint64_t b = off[l], e = off[l + 1];
        double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint64_t i = b + lane; i < e; i += 64u) {
            const uint32_t ib = id_base_at(b16, b32, i);
            const double pb = prob_b[ib >> 2], pa = 1 - pb;  // :24 weights by prob[group id]; :62-65
            const uint32_t base = ib & 3u;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                w[k] += base == k ? pa : 0.0;
                w[4 + k] += base == k ? pb : 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            for (int o = 32; o > 0; o >>= 1) w[k] += __shfl_down(w[k], o);
        if (lane == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double *c = w + 4 * h;
                double s = c[0] + c[1] + c[2] + c[3];  // :27
                if (s == 0) {                          // :28-30
                    for (int k = 0; k < 4; ++k) centres[(size_t)l * 8 + 4 * h + k] = log(0.25);
                    continue;
                }
                for (int k = 0; k < 4; ++k) c[k] = c[k] / s > theta ? c[k] / s : theta;  // :31-32
                s = c[0] + c[1] + c[2] + c[3];                                           // :34
                for (int k = 0; k < 4; ++k) centres[(size_t)l * 8 + 4 * h + k] = log(c[k] / s);  // :35-36
            }
        }
    }
}

// ll_a[cell] += sum over the cell's entries of centre_a[locus][base], same for b (:77-80, :142-145)
__global__ __launch_bounds__(256) void k_em_cell_sums(const uint32_t *cell_off, const uint32_t *val, uint32_t n_cells,
                                                     const double *centres, double *ll_a, double *ll_b,
                                                     const Flags *flags) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cell = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (cell >= n_cells || flags->done) return;
    double sa = 0.0, sb = 0.0;
    for (uint32_t i = cell_off[cell] + lane; i < cell_off[cell + 1]; i += 64u) {
        const uint32_t v = val[i];
        const double *c = centres + (size_t)(v >> 2) * 8 + (v & 3u);
        sa += c[0];
        sb += c[4];
    }
    for (int o = 32; o > 0; o >>= 1) {
        sa += __shfl_down(sa, o);
        sb += __shfl_down(sb, o);
    }
    if (lane == 0) {
        ll_a[cell] += sa;
        ll_b[cell] += sb;
    }
}

// expectation_step (:100-123), one workgroup
__global__ __launch_bounds__(1024) void k_em_estep(uint32_t n_cells, const double *ll_a, const double *ll_b,
                                                  double *prob_b, Flags *flags) {
    if (flags->done) return;
    __shared__ double part[1024];
    __shared__ int moved;
    double sum = 0.0;
    for (uint32_t i = threadIdx.x; i < n_cells; i += 1024u) sum += prob_b[i];
    part[threadIdx.x] = sum;
    if (threadIdx.x == 0) moved = 0;
    __syncthreads();
    for (uint32_t half = 512; half > 0; half >>= 1) {
        if (threadIdx.x < half) part[threadIdx.x] += part[threadIdx.x + half];
        __syncthreads();
    }
    const double prior_b = part[0] / n_cells, prior_a = 1 - prior_b;  // :109-110
    for (uint32_t i = threadIdx.x; i < n_cells; i += 1024u) {
        const double d = fmin(fmax(ll_b[i] - ll_a[i], -100.), 100.);  // :115
        const double odds = exp(d);
        const double prob = 1 - 1 / (1 + odds * prior_b / prior_a);   // :116
        if (!(fabs(prob - prob_b[i]) < 1e-2)) moved = 1;              // :117
        prob_b[i] = prob;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        flags->done = moved ? 0u : 1u;
        flags->iterations += 1u;
    }
}

int bits_for(uint32_t max_value) {
    int b = 1;
    while (b < 32 && (max_value >> b) != 0) ++b;
    return b;
}

int refine(int device_id, const uint64_t *d_off, uint32_t n_loci, uint64_t n_entries, const uint16_t *d_b16,
           const uint32_t *d_b32, const uint32_t *d_id_to_pos, uint32_t n_groups, double theta, double *d_prob,
           uint32_t n_cells, uint32_t max_iterations, uint32_t *iterations, hipStream_t stream) {
    if (!d_prob || (!d_off && n_loci)) return secedo::api_fail(SECEDO_E_INVALID_ARG, "null argument");
    if ((d_b16 != nullptr) == (d_b32 != nullptr) && n_entries)
        return secedo::api_fail(SECEDO_E_INVALID_ARG, "exactly one of id_base16 / id_base32 must be given");
    if (n_cells == 0) return secedo::api_fail(SECEDO_E_INVALID_ARG, "prob_cluster_b is empty");
    if (n_entries >= (1ull << 31) || n_loci >= (1u << 30))
        return secedo::api_fail(SECEDO_E_LIMIT, "pileup too large for the EM refinement (2^31 entries, 2^30 loci)");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return secedo::api_fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the EM refinement has no CPU fallback");
    if (device_id < 0 || device_id >= n_dev) return secedo::api_fail(SECEDO_E_NO_DEVICE, "device id out of range");
    EM_TRY(hipSetDevice(device_id));
    if (max_iterations == 0) max_iterations = 1000;
    const uint32_t E = (uint32_t)n_entries;

    // sizes first (the sort's temporary storage is a query), then one allocation carved up
    size_t sort_tmp = 0;
    if (E)
        EM_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)E, 0,
                                                  bits_for(n_cells - 1), stream));
    size_t total = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = total;
        total += (bytes + 255) / 256 * 256;
        return at;
    };
    const size_t o_key_a = carve((size_t)E * 4), o_key_b = carve((size_t)E * 4), o_val_a = carve((size_t)E * 4),
                 o_val_b = carve((size_t)E * 4), o_cub = carve(sort_tmp), o_cell_off = carve(((size_t)n_cells + 1) * 4),
                 o_centres = carve((size_t)std::max(n_loci, 1u) * 8 * sizeof(double)),
                 o_ll = carve((size_t)n_cells * 2 * sizeof(double)), o_flags = carve(sizeof(Flags));
    ArenaLease lease(device_id);
    EM_TRY(lease.a->ensure(total));
    unsigned char *base = static_cast<unsigned char *>(lease.a->p);
    uint32_t *key_a = reinterpret_cast<uint32_t *>(base + o_key_a), *key_b = reinterpret_cast<uint32_t *>(base + o_key_b);
    uint32_t *val_a = reinterpret_cast<uint32_t *>(base + o_val_a), *val_b = reinterpret_cast<uint32_t *>(base + o_val_b);
    uint32_t *cell_off = reinterpret_cast<uint32_t *>(base + o_cell_off);
    double *centres = reinterpret_cast<double *>(base + o_centres);
    Flags *flags = reinterpret_cast<Flags *>(base + o_flags);
    double *ll_a = reinterpret_cast<double *>(base + o_ll), *ll_b = ll_a + n_cells;
    EM_TRY(hipMemsetAsync(flags, 0, sizeof(Flags), stream));
    EM_TRY(hipMemsetAsync(ll_a, 0, (size_t)n_cells * 2 * sizeof(double), stream));  // :130-131
    const uint32_t locus_grid = std::max(1u, std::min((n_loci + 3u) / 4u, 16384u));
    const uint32_t *sorted_val = val_a;
    if (E) {
        hipLaunchKernelGGL(k_em_keys, dim3(locus_grid), dim3(256), 0, stream, d_off, n_loci, d_b16, d_b32, d_id_to_pos,
                           n_groups, n_cells, key_a, val_a, flags);
        EM_TRY(hipcub::DeviceRadixSort::SortPairs(base + o_cub, sort_tmp, key_a, key_b, val_a, val_b, (int)E, 0,
                                                  bits_for(n_cells - 1), stream));
        sorted_val = val_b;
    }
    hipLaunchKernelGGL(k_em_cell_offsets, dim3((n_cells + 256) / 256), dim3(256), 0, stream, key_b, E, n_cells, cell_off);
    Flags h{};
    EM_TRY(hipMemcpyAsync(&h, flags, sizeof(h), hipMemcpyDeviceToHost, stream));
    EM_TRY(hipStreamSynchronize(stream));
    if (h.error == 1)
        return secedo::api_fail(SECEDO_E_INVALID_ARG,
                                "a group id is >= n_cells: the reference indexes prob_cluster_b with the group id "
                                "(expectation_maximization.cpp:24) and would read out of bounds");
    if (h.error == 2)
        return secedo::api_fail(SECEDO_E_INVALID_ARG, "a group id is outside id_to_pos or maps outside prob_cluster_b");

    // Iterations are launched four at a time: a kernel returns at once when an earlier E-step of its
    // batch has settled, so the host reads the flags (a synchronisation, ~40 us) once per batch.
    uint32_t it = 0;
    for (;;) {
        if (it == max_iterations)
            return secedo::api_fail(SECEDO_E_LIMIT, "the EM refinement did not settle within max_iterations");
        const uint32_t batch = std::min(4u, max_iterations - it);
        for (uint32_t k = 0; k < batch; ++k) {
            if (n_loci)
                hipLaunchKernelGGL(k_em_centres, dim3(locus_grid), dim3(256), 0, stream, d_off, n_loci, d_b16, d_b32,
                                   d_prob, theta, centres, flags);
            hipLaunchKernelGGL(k_em_cell_sums, dim3((n_cells + 3u) / 4u), dim3(256), 0, stream, cell_off, sorted_val,
                               n_cells, centres, ll_a, ll_b, flags);
            hipLaunchKernelGGL(k_em_estep, dim3(1), dim3(1024), 0, stream, n_cells, ll_a, ll_b, d_prob, flags);
        }
        EM_TRY(hipMemcpyAsync(&h, flags, sizeof(h), hipMemcpyDeviceToHost, stream));
        EM_TRY(hipStreamSynchronize(stream));
        it = h.iterations;
        if (h.done) break;
    }
    EM_TRY(hipGetLastError());
    if (iterations) *iterations = it;
    return SECEDO_OK;
}

}  // namespace

namespace secedo {
void em_release_cache() {
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    for (auto &slot : g_arenas)
        if (!slot.second.busy) slot.second.release();
}
}  // namespace secedo

extern "C" {

int secedo_em_refine_device(int device_id, const uint64_t *d_locus_entry_off, uint32_t n_loci, uint64_t n_entries,
                            const uint16_t *d_id_base16, const uint32_t *d_id_base32, const uint32_t *d_id_to_pos,
                            uint32_t n_groups, double theta, double *d_prob_cluster_b, uint32_t n_cells,
                            uint32_t max_iterations, uint32_t *iterations, void *stream) {
    return refine(device_id, d_locus_entry_off, n_loci, n_entries, d_id_base16, d_id_base32, d_id_to_pos, n_groups,
                  theta, d_prob_cluster_b, n_cells, max_iterations, iterations, static_cast<hipStream_t>(stream));
}

int secedo_em_refine(int device_id, const uint64_t *locus_entry_off, uint32_t n_loci, const uint16_t *id_base16,
                     const uint32_t *id_base32, const uint32_t *id_to_pos, uint32_t n_groups, double theta,
                     double *prob_cluster_b, uint32_t n_cells, uint32_t max_iterations, uint32_t *iterations) {
    if (!prob_cluster_b || (!locus_entry_off && n_loci)) return secedo::api_fail(SECEDO_E_INVALID_ARG, "null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return secedo::api_fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the EM refinement has no CPU fallback");
    if (device_id < 0 || device_id >= n_dev) return secedo::api_fail(SECEDO_E_NO_DEVICE, "device id out of range");
    EM_TRY(hipSetDevice(device_id));
    const uint64_t E = n_loci ? locus_entry_off[n_loci] : 0;
    if (E && (id_base16 != nullptr) == (id_base32 != nullptr))
        return secedo::api_fail(SECEDO_E_INVALID_ARG, "exactly one of id_base16 / id_base32 must be given");
    Buf off, b, i2p, prob;
    EM_TRY(off.alloc(((size_t)n_loci + 1) * 8));
    EM_TRY(b.alloc(E * (id_base16 ? 2 : 4)));
    EM_TRY(i2p.alloc((size_t)n_groups * 4));
    EM_TRY(prob.alloc((size_t)n_cells * 8));
    if (n_loci) EM_TRY(hipMemcpy(off.p, locus_entry_off, ((size_t)n_loci + 1) * 8, hipMemcpyHostToDevice));
    if (E) EM_TRY(hipMemcpy(b.p, id_base16 ? (const void *)id_base16 : (const void *)id_base32,
                            E * (id_base16 ? 2 : 4), hipMemcpyHostToDevice));
    if (n_groups) EM_TRY(hipMemcpy(i2p.p, id_to_pos, (size_t)n_groups * 4, hipMemcpyHostToDevice));
    if (n_cells) EM_TRY(hipMemcpy(prob.p, prob_cluster_b, (size_t)n_cells * 8, hipMemcpyHostToDevice));
    const int rc = refine(device_id, off.as<uint64_t>(), n_loci, E, id_base16 ? b.as<uint16_t>() : nullptr,
                          id_base16 ? nullptr : b.as<uint32_t>(), i2p.as<uint32_t>(), n_groups, theta,
                          prob.as<double>(), n_cells, max_iterations, iterations, nullptr);
    if (rc) return rc;
    EM_TRY(hipMemcpy(prob_cluster_b, prob.p, (size_t)n_cells * 8, hipMemcpyDeviceToHost));
    return SECEDO_OK;
}

}  // extern "C"
