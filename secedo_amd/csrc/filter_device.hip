// filter_device.hip -- the locus filter on the GPU (reference: util/is_significant.cpp:149-193
// Filter::filter, :78-138 Filter::is_significant), SURVEY.md section 8f rank 2.
//
// Streaming, HBM-bound: every entry is read once (4 B read id + 2/4 B id|base) and the kept ones
// are written once. One wave per locus:
//   k_decide   base counts of the entries whose group is in the sub-cluster (id_to_pos != NO_POS),
//              the reference's integer pre-tests, then the log-likelihood statistic in fp64. A locus
//              whose statistic lies within 1e-9 of its threshold is marked UNSURE and decided by the
//              host with the C library's pow/log, so that the decision equals the reference's even
//              where the device's libm differs in the last bit.
//   (scan)     kept loci -> output locus index; kept entries -> output entry offsets
//   k_compact  stable compaction of the kept loci's in-cluster entries
#include "filter_device.hpp"

#include "filter_host.hpp"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

namespace secedo {

namespace {

constexpr int TPB = 256;
constexpr uint32_t UNSURE = 2;

struct Thresholds {
    double k[20];
};

struct FilterIn {
    const uint32_t *chr_locus_off;
    uint32_t n_chr;
    const uint32_t *locus_pos;
    const uint64_t *locus_entry_off;
    const uint32_t *read_ids;
    const uint16_t *id_base16;
    const uint32_t *id_base32;
    const uint32_t *id_to_pos;
    uint32_t n_groups;
    uint32_t n_loci;
    uint64_t n_entries;
    __device__ __forceinline__ uint32_t id_base(uint64_t e) const {
        return id_base16 ? (uint32_t)id_base16[e] : id_base32[e];
    }
};

__device__ __forceinline__ void sort4(uint32_t c[4]) {
#define CSWAP(a, b) { const uint32_t lo = min(c[a], c[b]), hi = max(c[a], c[b]); c[a] = lo; c[b] = hi; }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
}

// in_count[l]: entries of the sub-cluster; counts4[l]: the four base counts. One wave per locus.
// (The verdict is k_verdict's, a THREAD per locus: taken here by lane 0 of the locus' wave, its dozen fp64 logarithms
// and powers ran on one lane in 64 and were most of the 140 us this kernel took on C3.)
__global__ __launch_bounds__(TPB) void k_decide(FilterIn in, uint32_t *in_count, uint4 *counts4) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    for (uint32_t l = wave; l < in.n_loci; l += n_waves) {
        const uint64_t e0 = in.locus_entry_off[l], e1 = in.locus_entry_off[l + 1];
        uint32_t c[4] = {0, 0, 0, 0};
        if (in.id_base16 && (reinterpret_cast<uintptr_t>(in.id_base16) & 7u) == 0u) {
            // four 2-byte entries per lane and load: a wave takes 256 entries of the locus at a time (2 bytes per lane,
            // 64 entries at a time, the kernel ran at 1.1 TB/s of its one input stream)
            for (uint64_t a = e0 & ~3ull; a < e1; a += 256) {
                const uint64_t e = a + (uint64_t)lane * 4u;
                if (e >= e1) continue;
                uint32_t v4[4];
                if (e + 4u <= in.n_entries) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(in.id_base16 + e);
                    v4[0] = w.x & 0xFFFFu, v4[1] = w.x >> 16, v4[2] = w.y & 0xFFFFu, v4[3] = w.y >> 16;
                } else {  // (the last entries of the pileup: nothing is read behind the array)
#pragma unroll
                    for (int u = 0; u < 4; ++u) v4[u] = e + (uint64_t)u < in.n_entries ? in.id_base16[e + (uint64_t)u] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (e + (uint64_t)u < e0 || e + (uint64_t)u >= e1) continue;
                    const uint32_t g = v4[u] >> 2;
                    if (g < in.n_groups && in.id_to_pos[g] != kNoPos) c[v4[u] & 3u]++;
                }
            }
        } else {
            for (uint64_t e = e0 + lane; e < e1; e += 64) {
                const uint32_t v = in.id_base(e);
                const uint32_t g = v >> 2;
                if (g < in.n_groups && in.id_to_pos[g] != kNoPos) c[v & 3u]++;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            for (int off = 32; off > 0; off >>= 1) c[b] += __shfl_down(c[b], off);
        }
        if (lane != 0) continue;
        in_count[l] = c[0] + c[1] + c[2] + c[3];
        counts4[l] = make_uint4(c[0], c[1], c[2], c[3]);
    }
}

// decision[l]: 0 drop, 1 keep, 2 unsure (host decides)
__global__ __launch_bounds__(TPB) void k_verdict(uint32_t n_loci, double theta, Thresholds th, const uint4 *counts4,
                                                uint32_t *decision, uint32_t *n_unsure) {
    for (uint32_t l = blockIdx.x * TPB + threadIdx.x; l < n_loci; l += gridDim.x * TPB) {
        const uint4 c4 = counts4[l];
        uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
        const uint32_t coverage = c[0] + c[1] + c[2] + c[3];
        uint32_t verdict = 0;
        if (coverage > 65535u) {
            verdict = UNSURE;  // the reference counts in uint16 and wraps: leave it to the host
        } else if (coverage >= 2) {
            sort4(c);
            if (c[2] != 0 && c[2] + c[1] + c[0] >= 5 && !((double)c[3] < 1.5 * (double)c[2])) {
                double col = rint((double)coverage / 10.) - 1.;
                col = fmin(fmax(col, 0.), 19.);
                const double k = th.k[(uint32_t)col];
                const double hetero_prior = 0.0005, mut_prior = 1e-6;
                const double homo_prior = 1 - hetero_prior - mut_prior;
                double log_homozygous = c[3] * log(1 - theta) + (coverage - c[3]) * log(theta / 3);
                log_homozygous += log(1. / 4);
                log_homozygous += log(hetero_prior);
                const double all_c1 = homo_prior * pow(1 - theta, (double)c[3]) * pow(theta / 3, (double)(coverage - c[3]));
                const double hetero = hetero_prior * pow(0.5 - theta / 3, (double)(c[3] + c[2]))
                        * pow(theta / 3, (double)(c[0] + c[1]));
                const double homo_som = homo_prior * mut_prior * pow(0.75 - 2 * theta / 3, (double)c[3])
                        * pow(0.25, (double)c[2]) * pow(theta / 3, (double)(c[0] + c[1]));
                const double hetero_som = hetero_prior * mut_prior * pow(0.5 - theta, (double)c[3])
                        * pow(0.25, (double)(c[1] + c[2])) * pow(theta / 3, (double)c[0]);
                const double two_som = hetero_prior * mut_prior * mut_prior * pow(1 - theta, (double)coverage);
                const double s = log_homozygous - log(all_c1 + hetero + homo_som + hetero_som + two_som);
                if (!(fabs(s - k) > 1e-9 * fmax(1., fabs(k)))) verdict = UNSURE;  // also catches NaN
                else verdict = s < k ? 1u : 0u;
            }
        }
        decision[l] = verdict;
        if (verdict == UNSURE) atomicAdd(n_unsure, 1u);
    }
}

__global__ void k_weights(const uint32_t *decision, const uint32_t *in_count, uint32_t n, uint32_t *keep,
                          unsigned long long *kept_entries) {
    for (uint32_t l = blockIdx.x * TPB + threadIdx.x; l < n; l += gridDim.x * TPB) {
        const uint32_t k = decision[l] == 1u ? 1u : 0u;
        keep[l] = k;
        kept_entries[l] = k ? in_count[l] : 0ull;
    }
}

// stable compaction of the in-cluster entries of every kept locus; one wave per locus
template <class IdBase>
__global__ __launch_bounds__(TPB) void k_compact(FilterIn in, const IdBase *id_base, const uint32_t *keep,
                                                const uint32_t *locus_rank, const unsigned long long *entry_rank,
                                                uint32_t *out_pos, unsigned long long *out_off,
                                                uint32_t *out_rid, IdBase *out_idb) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    for (uint32_t l = wave; l < in.n_loci; l += n_waves) {
        if (!keep[l]) continue;
        const uint64_t e0 = in.locus_entry_off[l], e1 = in.locus_entry_off[l + 1];
        unsigned long long dst = entry_rank[l];
        if (lane == 0) {
            out_pos[locus_rank[l]] = in.locus_pos[l];
            out_off[locus_rank[l]] = dst;
        }
        for (uint64_t base = e0; base < e1; base += 64) {
            const uint64_t e = base + lane;
            bool in_cluster = false;
            IdBase v = 0;
            if (e < e1) {
                v = id_base[e];
                const uint32_t g = (uint32_t)v >> 2;
                in_cluster = g < in.n_groups && in.id_to_pos[g] != kNoPos;
            }
            const unsigned long long mask = __ballot(in_cluster);
            if (in_cluster) {
                const uint32_t before = __popcll(mask & ((1ull << lane) - 1ull));
                out_rid[dst + before] = in.read_ids[e];
                out_idb[dst + before] = v;
            }
            dst += __popcll(mask);
        }
    }
}

__global__ void k_chr_offsets(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_rank,
                              uint32_t n_loci, uint32_t total_kept, uint32_t *out_chr_locus_off,
                              const unsigned long long *entry_rank, unsigned long long *out_locus_entry_off) {
    if (blockIdx.x == 0 && threadIdx.x == 0) out_locus_entry_off[total_kept] = entry_rank[n_loci];
    for (uint32_t c = blockIdx.x * TPB + threadIdx.x; c <= n_chr; c += gridDim.x * TPB) {
        const uint32_t l = chr_locus_off[c];
        out_chr_locus_off[c] = l < n_loci ? locus_rank[l] : total_kept;
    }
}

inline uint32_t blocks_for(uint64_t n) {
    return static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((n + TPB - 1) / TPB, 1u << 16)));
}

#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e__); \
    } while (0)

}  // namespace

std::string filter_device(const DeviceFlatPileup &in, double theta, uint32_t cell_proportion,
                          hipStream_t stream, FilterWorkspace *ws, const FilterOut &out, uint64_t *n_loci_out,
                          uint64_t *n_entries_out, double *avg_coverage) {
    if (cell_proportion > 4) return "cell_proportion must be in [0, 4]";
    if ((in.id_base16 != nullptr) == (in.id_base32 != nullptr))
        return "exactly one of id_base16 / id_base32 must be given";
    const uint32_t L = in.n_loci;
    *n_loci_out = 0;
    *n_entries_out = 0;
    *avg_coverage = 0;
    if (L == 0) {
        HIP_OK(hipMemsetAsync(out.chr_locus_off, 0, ((size_t)in.n_chr + 1) * 4, stream));
        HIP_OK(hipMemsetAsync(out.locus_entry_off, 0, 8, stream));
        return std::string();
    }
    FilterIn fin{in.chr_locus_off, in.n_chr, in.locus_pos, in.locus_entry_off, in.read_ids, in.id_base16,
                 in.id_base32, in.group_id_to_pos, in.n_groups, L, in.n_entries};
    Thresholds th;
    for (int i = 0; i < 20; ++i) th.k[i] = kSignificanceThresholds[cell_proportion][i];

    // workspace: decision[L] | in_count[L] | keep[L] | locus_rank[L+1] | counts4[L] | kept[L+1] u64 | rank[L+1] u64 | scalars
    HIP_OK(ws->a.ensure(((size_t)L + 1) * (4 * 4 + 16 + 8 + 8) + 64));
    unsigned char *base = ws->a.as<unsigned char>();
    uint4 *counts4 = reinterpret_cast<uint4 *>(base);
    unsigned long long *kept_entries = reinterpret_cast<unsigned long long *>(counts4 + (L + 1));
    unsigned long long *entry_rank = kept_entries + (L + 1);
    uint32_t *decision = reinterpret_cast<uint32_t *>(entry_rank + (L + 1));
    uint32_t *in_count = decision + (L + 1), *keep = in_count + (L + 1), *locus_rank = keep + (L + 1);
    uint32_t *n_unsure = locus_rank + (L + 1);
    size_t need = 0, most = 0;
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, keep, locus_rank, (int)L + 1, stream));
    most = std::max(most, need);
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, kept_entries, entry_rank, (int)L + 1, stream));
    most = std::max(most, need);
    HIP_OK(ws->b.ensure(most + 256));

    HIP_OK(hipMemsetAsync(n_unsure, 0, 4, stream));
    const uint32_t wave_grid = static_cast<uint32_t>(std::min<uint64_t>(((uint64_t)L * 64 + TPB - 1) / TPB, 1u << 15));
    hipLaunchKernelGGL(k_decide, dim3(wave_grid), dim3(TPB), 0, stream, fin, in_count, counts4);
    hipLaunchKernelGGL(k_verdict, dim3(blocks_for(L)), dim3(TPB), 0, stream, L, theta, th, counts4, decision, n_unsure);
    // Ranks of the kept loci and entries. The verdicts are assumed final (a locus whose statistic touches
    // its threshold is rare): weights, scans and ONE read-back of {unsure, kept loci, kept entries}; only
    // if a locus was unsure is it decided on the host and the ranking redone.
    uint32_t total_loci = 0;
    unsigned long long total_entries = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        hipLaunchKernelGGL(k_weights, dim3(blocks_for(L)), dim3(TPB), 0, stream, decision, in_count, L, keep, kept_entries);
        HIP_OK(hipMemsetAsync(keep + L, 0, 4, stream));
        HIP_OK(hipMemsetAsync(kept_entries + L, 0, 8, stream));
        size_t cap = ws->b.bytes;
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(ws->b.p, cap, keep, locus_rank, (int)L + 1, stream));
        cap = ws->b.bytes;
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(ws->b.p, cap, kept_entries, entry_rank, (int)L + 1, stream));
        uint32_t h_unsure = 0;
        if (attempt == 0) HIP_OK(hipMemcpyAsync(&h_unsure, n_unsure, 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(&total_loci, locus_rank + L, 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(&total_entries, entry_rank + L, 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        if (!h_unsure) break;
        // the few loci whose statistic touches its threshold: decided with the C library, like the reference
        std::vector<uint32_t> dec(L);
        std::vector<uint4> cnt(L);
        HIP_OK(hipMemcpy(dec.data(), decision, (size_t)L * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(cnt.data(), counts4, (size_t)L * 16, hipMemcpyDeviceToHost));
        for (uint32_t l = 0; l < L; ++l) {
            if (dec[l] != UNSURE) continue;
            // the reference counts in uint16 (std::array<uint16_t, 4>): wrap like it does
            const uint16_t c[4] = {(uint16_t)cnt[l].x, (uint16_t)cnt[l].y, (uint16_t)cnt[l].z, (uint16_t)cnt[l].w};
            dec[l] = is_significant(c, theta, cell_proportion) ? 1u : 0u;
        }
        HIP_OK(hipMemcpy(decision, dec.data(), (size_t)L * 4, hipMemcpyHostToDevice));
    }
    if (in.id_base16) {
        hipLaunchKernelGGL((k_compact<uint16_t>), dim3(wave_grid), dim3(TPB), 0, stream, fin, in.id_base16, keep,
                           locus_rank, entry_rank, out.locus_pos, reinterpret_cast<unsigned long long *>(out.locus_entry_off),
                           out.read_ids, static_cast<uint16_t *>(out.id_base));
    } else {
        hipLaunchKernelGGL((k_compact<uint32_t>), dim3(wave_grid), dim3(TPB), 0, stream, fin, in.id_base32, keep,
                           locus_rank, entry_rank, out.locus_pos, reinterpret_cast<unsigned long long *>(out.locus_entry_off),
                           out.read_ids, static_cast<uint32_t *>(out.id_base));
    }
    // ... and the closing offset of the last kept locus
    hipLaunchKernelGGL(k_chr_offsets, dim3(1), dim3(TPB), 0, stream, in.chr_locus_off, in.n_chr, locus_rank, L,
                       total_loci, out.chr_locus_off, entry_rank,
                       reinterpret_cast<unsigned long long *>(out.locus_entry_off));
    HIP_OK(hipStreamSynchronize(stream));
    *n_loci_out = total_loci;
    *n_entries_out = total_entries;
    // the reference sums the coverage in uint32 (util/is_significant.cpp:186-188)
    *avg_coverage = total_loci == 0 ? 0.0 : (double)(uint32_t)total_entries / (double)total_loci;
    return std::string();
}

}  // namespace secedo
