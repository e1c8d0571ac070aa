// pack_device.hip -- device-side packing of the raw flat pileup (see pack_device.hpp).
//
// Pipeline (every step a kernel or a hipCUB scan on `stream`; 3 scalar read-backs):
//   1  entry -> locus; entries grouped by (chromosome, read id) in pileup order. Sparse loci (most reads have one
//      entry): every entry stores its number in its id's slot and checks whether the slot still holds it -- the
//      entries of repeated ids ("M entries") are flagged, numbered, compacted and grouped by the entry that won
//      their slot; the others are their reads. Otherwise: histogram over the dense id space, scan, scatter with an
//      atomic cursor, rank inside the (short) group; the hipCUB radix sort when the ids are sparse or a group is
//      too long for that
//   2  duplicate-position rule per (read, locus) group (reference: similarity_matrix.cpp:387-395);
//      reads = runs of equal key, cut further where a flush erases a read that outlives
//      max_fragment_length (:368-371, :379-382; k_split_update, iterated with step 4)
//   3  kept entries -> per-read lists (CSR), multi-locus statistics
//   4  first-appearance rank of every read, completed-prefix count per locus, flush chain per
//      chromosome (reference :348-373) -> number of flushed reads F_c -> tail flags (:407-408)
//   5  kept entries grouped by (cell block, locus, cell) -- per-locus LDS histograms, scan, placement,
//      in-group ranking (or the radix sort) --, block offsets, pair bound, locus ranges, entry records
//      (window masks) written at their final position
#include "pack_device.hpp"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace secedo {

DeviceArena::~DeviceArena() {
    if (p) (void)hipFree(p);
}

DevicePacked::~DevicePacked() {
    if (side) (void)hipStreamDestroy(side);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_offsets) (void)hipEventDestroy(ev_offsets);
    if (ev_flush) (void)hipEventDestroy(ev_flush);
    if (mailbox) (void)hipHostFree(mailbox);
}

// SECEDO_POISON (a debugging aid): 1 fills every new allocation with 0xA5 bytes, 2 also refills the packing's
// scratch and outputs at the start of every prepare -- a kernel that reads a word nobody wrote then reads
// 0xA5A5A5A5 every time instead of whatever the allocation happened to hold (tools/poison_run.sh).
int poison_level() {
    static const int level = [] {
        const char *e = std::getenv("SECEDO_POISON");
        return e ? std::atoi(e) : 0;
    }();
    return level;
}

hipError_t DeviceArena::ensure(size_t n) {
    if (n <= bytes && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    const size_t want = n ? n + n / 8 + 256 : 256;  // head room: sizes vary a little from call to call
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) bytes = want;
    if (e == hipSuccess && poison_level() >= 1) e = hipMemset(p, 0xA5, want);
    return e;
}

namespace {

constexpr int TPB = 256;

inline uint32_t blocks_for(uint64_t n) {
    return static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((n + TPB - 1) / TPB, 1u << 16)));
}

struct Scalars {
    uint32_t error;       // 1 group id outside group_id_to_pos, 2 row outside the matrix, 3 positions not increasing
    uint32_t need_host;
    uint32_t num_ranges;
    uint32_t max_read_id;
    uint32_t regroup;     // a group too long for the in-group ranking: redo with the radix sorts
    uint32_t long_reads;  // some id's entries span >= max_fragment_length: flushes may split it (k_split_update)
    uint32_t split_changed;
    uint32_t id_exceeded;  // the id space is larger than the caller assumed (the size of the previous call)
    uint32_t max_read_entries;  // kept entries of the longest read id run
    uint32_t caps_wrong;        // the locus ranges were cut for the count tile, and the pair bound forbids it
    uint32_t n_multi_id;        // entries whose read id occurs more than once in its chromosome (k_compact_m)
    uint32_t reads_total;       // reads of the whole pileup (k_arank_m)
    uint32_t n_wide;            // kept entries whose read reaches beyond their 8-locus windows (k_m_records / k_records)
    unsigned long long id_space;  // sum over chromosomes of (largest - smallest read id + 1)
    unsigned long long multi_entries;
    unsigned long long pair_bound;
};

// Scalar read-backs. hipMemcpyAsync + hipStreamSynchronize leaves the GPU idle for 30-40 us per
// read-back (interrupt, wake-up, the first launches after it); instead a one-lane kernel publishes the
// scalars into pinned, coherent (fine-grained: visible to the host while the stream is still running,
// whatever HIP_HOST_COHERENT says) host memory and the host polls the sequence word next to them.
struct Mailbox {
    Scalars sc;
    unsigned long long totals;
    unsigned long long seq;
};

__global__ void k_publish(const Scalars *sc, const unsigned long long *totals, Mailbox *box, unsigned long long seq) {
    if (threadIdx.x == 0) {
        box->sc = *sc;
        box->totals = totals ? *totals : 0ull;
        __hip_atomic_store(&box->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t read_scalars(DevicePacked &pk, hipStream_t stream, const Scalars *sc, const unsigned long long *totals,
                        Scalars *out, unsigned long long *out_totals) {
    static const bool plain = [] {
        const char *e = std::getenv("SECEDO_PACK_READBACK");
        return e && std::strcmp(e, "memcpy") == 0;
    }();
    hipError_t e;
    if (!plain && !pk.mailbox && !pk.mailbox_failed) {
        if (hipHostMalloc(&pk.mailbox, sizeof(Mailbox), hipHostMallocCoherent) == hipSuccess) {
            std::memset(pk.mailbox, 0, sizeof(Mailbox));
        } else {  // no pinned memory to be had: the plain copies below
            (void)hipGetLastError();
            pk.mailbox = nullptr;
            pk.mailbox_failed = true;
        }
    }
    if (!plain && pk.mailbox) {
        Mailbox *box = static_cast<Mailbox *>(pk.mailbox);
        const unsigned long long seq = ++pk.mailbox_seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, stream, sc, totals, box, seq);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // poll; every few thousand polls ask the runtime whether the stream died or drained without
        // the word arriving (then the plain copy below reports what happened)
        bool arrived = false;
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(&box->seq, __ATOMIC_ACQUIRE) == seq) {
                arrived = true;
                break;
            }
            if ((spins & 0xFFFu) == 0xFFFu) {
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipErrorNotReady) {
                    arrived = __atomic_load_n(&box->seq, __ATOMIC_ACQUIRE) == seq;
                    if (q != hipSuccess) return q;
                    break;
                }
            }
        }
        if (arrived) {
            *out = box->sc;
            if (out_totals) *out_totals = box->totals;
            return hipSuccess;
        }
    }
    if (out_totals) {
        *out_totals = 0;
        if (totals && (e = hipMemcpyAsync(out_totals, totals, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    }
    if ((e = hipMemcpyAsync(out, sc, sizeof(*out), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    return hipStreamSynchronize(stream);
}

// SECEDO_PACK_TRACE=1: host-side time stamps of the packing's launch sequence on stderr
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    HostTrace() : on(std::getenv("SECEDO_PACK_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) const {
        if (on)
            std::fprintf(stderr, "[pack-trace] %8.1f us  %s\n",
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what);
    }
};

// Grouping without a sort (entries by read id, kept entries by (cell block, locus)): histogram of
// the group sizes, exclusive scan, scatter with an atomic cursor (arbitrary order inside a group),
// then every element finds its rank inside its group by scanning the group -- groups are short
// (a read covers a few loci; a cell block has a few entries at a locus), so this costs a few cached
// reads per element where a radix sort costs 4-7 passes over the data. The result is the order a
// stable sort gives. A group longer than kRankScanLimit raises Scalars::regroup and the caller
// falls back to the radix sorts.
constexpr uint32_t kRankScanLimit = 8192;
constexpr int kNoRetry = 0, kRetryRadix = 1, kRetrySameScheme = 2;  // what an attempt asks of its caller
constexpr int kMaxSplitRounds = 32;  // rounds of k_split_update before the host emulation takes over
// the counting scheme for read ids needs a table over the id space: used while max id < factor * entries
constexpr uint32_t kIdSpaceFactor = 4;

struct Raw {  // by-value kernel argument: the raw pileup
    const uint32_t *chr_locus_off;
    uint32_t n_chr;
    const uint32_t *locus_pos;
    const uint64_t *locus_entry_off;
    const uint32_t *read_ids;
    const uint16_t *id_base16;
    const uint32_t *id_base32;
    const uint32_t *g2p;
    uint32_t n_groups;
    uint32_t n_loci;
    uint32_t n_entries;
    __device__ __forceinline__ uint32_t id_base(uint32_t e) const {
        return id_base16 ? (uint32_t)id_base16[e] : id_base32[e];
    }
};

// index of the last element <= x in a non-decreasing array a[0..n) with a[0] <= x  (u64 / u32)
template <class T>
__device__ __forceinline__ uint32_t last_le(const T *a, uint32_t n, T x) {
    uint32_t lo = 0, hi = n;  // invariant: a[lo] <= x, (hi == n or a[hi] > x)
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (a[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// entry -> locus, one wave per locus (coalesced; empty loci cost nothing); lane 0 also checks that the
// positions increase strictly inside a chromosome (the reference asserts it, similarity_matrix.cpp:398)
__global__ __launch_bounds__(TPB) void k_entry_locus(Raw in, uint32_t *entry_locus, Scalars *sc) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    for (uint32_t l = wave; l < in.n_loci; l += n_waves) {
        const uint32_t b = (uint32_t)in.locus_entry_off[l], e = (uint32_t)in.locus_entry_off[l + 1];
        for (uint32_t i = b + lane; i < e; i += 64u) entry_locus[i] = l;
        if (lane == 0u && l + 1 < in.n_loci && in.locus_pos[l + 1] <= in.locus_pos[l]) {
            // allowed only across a chromosome boundary
            const uint32_t c = last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, l);
            if (l + 1 < in.chr_locus_off[c + 1]) sc->error = 3;
        }
    }
}

// per chromosome the largest and the smallest read id (the latter as max of ~id, so that zero-filled
// memory is the neutral element). A workgroup owns a contiguous chunk of entries and cuts it at the
// chromosome boundaries: one pair of atomics per (workgroup, chromosome) -- same-address atomics
// are slow.
__global__ __launch_bounds__(TPB) void k_id_range(Raw in, const uint32_t *entry_locus, uint32_t *id_max,
                                                 uint32_t *id_negmin, int vec) {
    __shared__ uint32_t s_hi[TPB / 64], s_neg[TPB / 64];
    const uint32_t E = in.n_entries;
    const uint32_t chunk = (E + gridDim.x - 1) / gridDim.x;
    uint32_t cur = min(E, blockIdx.x * chunk);
    const uint32_t e1 = min(E, cur + chunk);
    if (cur >= e1) return;
    // (the locus of the first entry from k_entry_locus' table: a binary search over the loci here is 16
    // dependent reads in front of every workgroup's chunk and was most of this kernel's time)
    uint32_t c = last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, entry_locus[cur]);
    while (cur < e1) {
        // entries of chromosome c end where its last locus ends (skip chromosomes without entries)
        uint32_t c_end = (uint32_t)in.locus_entry_off[in.chr_locus_off[c + 1]];
        while (c_end <= cur && c + 1 < in.n_chr) {
            ++c;
            c_end = (uint32_t)in.locus_entry_off[in.chr_locus_off[c + 1]];
        }
        const uint32_t seg_end = (c + 1 < in.n_chr) ? min(e1, c_end) : e1;
        uint32_t hi = 0, neg = 0;
        uint32_t s_cur = cur, s_end = seg_end;  // what the scalar loop below is left with
        if (vec) {  // (the ids 16 bytes per lane, four loads in flight; the few before / behind the aligned part one by one)
            const uint32_t a = min(seg_end, (cur + 3u) & ~3u), b = max(a, seg_end & ~3u);
            if (threadIdx.x < a - cur) {
                const uint32_t id = in.read_ids[cur + threadIdx.x];
                hi = max(hi, id);
                neg = max(neg, ~id);
            }
            if (threadIdx.x < seg_end - b) {
                const uint32_t id = in.read_ids[b + threadIdx.x];
                hi = max(hi, id);
                neg = max(neg, ~id);
            }
            const uint4 *src = reinterpret_cast<const uint4 *>(in.read_ids);
            for (uint32_t q = a / 4u + threadIdx.x; q < b / 4u; q += TPB * 4) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = q + (uint32_t)u * TPB < b / 4u ? src[q + (uint32_t)u * TPB] : src[q];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    hi = max(max(hi, v[u].x), max(max(v[u].y, v[u].z), v[u].w));
                    neg = max(max(neg, ~v[u].x), max(max(~v[u].y, ~v[u].z), ~v[u].w));
                }
            }
            s_cur = s_end = seg_end;
        }
        // (eight loads of a thread in flight: one at a time, the kernel was 49 us of waiting on C3)
        for (uint32_t e = s_cur + threadIdx.x; e < s_end; e += TPB * 8) {
            uint32_t id[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t eu = e + (uint32_t)u * TPB;
                id[u] = eu < s_end ? in.read_ids[eu] : in.read_ids[e];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                hi = max(hi, id[u]);
                neg = max(neg, ~id[u]);
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            hi = max(hi, (uint32_t)__shfl_down(hi, off));
            neg = max(neg, (uint32_t)__shfl_down(neg, off));
        }
        __syncthreads();  // the previous segment's partials have been read
        if ((threadIdx.x & 63u) == 0) {
            s_hi[threadIdx.x >> 6] = hi;
            s_neg[threadIdx.x >> 6] = neg;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < TPB / 64; ++w) {
                hi = max(hi, s_hi[w]);
                neg = max(neg, s_neg[w]);
            }
            atomicMax(&id_max[c], hi);
            atomicMax(&id_negmin[c], neg);
        }
        cur = seg_end;
    }
}

// dense numbering of (chromosome, read id): id_base[c] + id - smallest id of c
__global__ void k_id_bases(uint32_t n_chr, const uint32_t *id_max, const uint32_t *id_negmin, uint32_t *id_base,
                           unsigned long long assumed_space, Scalars *sc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long sum = 0;
    uint32_t top = 0;
    for (uint32_t c = 0; c < n_chr; ++c) {
        id_base[c] = (uint32_t)min(sum, 0xFFFFFFFFull);
        const uint32_t lo = ~id_negmin[c];
        if ((id_max[c] | id_negmin[c]) != 0u) sum += (unsigned long long)(id_max[c] - lo) + 1ull;  // has entries
        top = max(top, id_max[c]);
    }
    id_base[n_chr] = (uint32_t)min(sum, 0xFFFFFFFFull);
    sc->id_space = sum;
    sc->max_read_id = top;
    // the caller may have sized the histogram from the previous call without waiting for this kernel
    if (assumed_space && sum > assumed_space) sc->id_exceeded = 1;
}

// radix path: sort key (chromosome, read id) with the read id in id_bits bits
__global__ void k_entry_keys(Raw in, const uint32_t *entry_locus, uint32_t id_bits, unsigned long long *key,
                             uint32_t *val) {
    for (uint32_t e = blockIdx.x * TPB + threadIdx.x; e < in.n_entries; e += gridDim.x * TPB) {
        const uint32_t c = last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, entry_locus[e]);
        key[e] = ((unsigned long long)c << id_bits) | in.read_ids[e];
        val[e] = e;
    }
}

// Per entry in sorted order: locus (30 bits) | base << 30. Written once with the one gather by entry
// index; the kernels of stages 2-4 then read it sequentially instead of gathering through sval.
constexpr uint32_t kSlocLocusMask = 0x3FFFFFFFu;
__device__ __forceinline__ uint32_t sloc_pack(uint32_t locus, uint32_t id_base) { return locus | (id_base << 30); }
__device__ __forceinline__ uint32_t sloc_locus(uint32_t v) { return v & kSlocLocusMask; }
__device__ __forceinline__ uint32_t sloc_base(uint32_t v) { return v >> 30; }

// radix path (the counting path writes sloc in k_id_rank)
__global__ void k_sorted_locus(Raw in, const uint32_t *sval, const uint32_t *entry_locus, uint32_t n, uint32_t *sloc) {
    for (uint32_t s = blockIdx.x * TPB + threadIdx.x; s < n; s += gridDim.x * TPB)
        sloc[s] = sloc_pack(entry_locus[sval[s]], in.id_base(sval[s]));
}

// The dense id of an entry: its chromosome (a search over the chromosomes' first loci), then two per-chromosome words.
// From LDS when the tables fit: from global memory the search is a chain of log2(C) dependent loads in front of
// whatever the kernel does with the id (k_id_store: 156 -> 117 us on C3).
constexpr uint32_t kChrLds = 1024;
struct ChrTables {
    uint32_t first[kChrLds + 1], base[kChrLds], min_id[kChrLds];
};
__device__ __forceinline__ bool chr_tables_load(const Raw &in, const uint32_t *id_base, const uint32_t *id_negmin,
                                                ChrTables &t) {  // (all threads; ends with a barrier)
    const bool in_lds = in.n_chr <= kChrLds;
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i <= in.n_chr; i += blockDim.x) t.first[i] = in.chr_locus_off[i];
        for (uint32_t i = threadIdx.x; i < in.n_chr; i += blockDim.x) {
            t.base[i] = id_base[i];
            t.min_id[i] = ~id_negmin[i];
        }
    }
    __syncthreads();
    return in_lds;
}
__device__ __forceinline__ uint32_t dense_id(const Raw &in, const uint32_t *id_base, const uint32_t *id_negmin,
                                             const ChrTables &t, bool in_lds, uint32_t locus, uint32_t id) {
    if (in_lds) {
        const uint32_t c = last_le<uint32_t>(t.first, in.n_chr + 1, locus);
        return t.base[c] + (id - t.min_id[c]);
    }
    const uint32_t c = last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, locus);
    return id_base[c] + (id - ~id_negmin[c]);
}

// counting path, entries by (chromosome, read id) through the dense numbering
__global__ void k_id_hist(Raw in, const uint32_t *entry_locus, const uint32_t *id_base, const uint32_t *id_negmin,
                          const Scalars *sc, uint32_t *dense, uint32_t *hist) {
    if (sc->id_exceeded) return;  // the table is too small: the caller starts over (k_id_rank)
    __shared__ ChrTables tables;
    const bool in_lds = chr_tables_load(in, id_base, id_negmin, tables);
    for (uint32_t e = blockIdx.x * TPB + threadIdx.x; e < in.n_entries; e += gridDim.x * TPB) {
        const uint32_t d = dense_id(in, id_base, id_negmin, tables, in_lds, entry_locus[e], in.read_ids[e]);
        dense[e] = d;
        atomicAdd(&hist[d], 1u);
    }
}

__global__ void k_id_scatter(const uint32_t *dense, uint32_t n, const uint32_t *id_off, const Scalars *sc,
                             uint32_t *hist, uint32_t *grouped) {
    if (sc->id_exceeded) return;
    for (uint32_t e = blockIdx.x * TPB + threadIdx.x; e < n; e += gridDim.x * TPB) {
        const uint32_t d = dense[e];
        grouped[id_off[d] + atomicSub(&hist[d], 1u) - 1u] = e;
    }
}

// rank inside the read's group = number of its entries with a smaller index: the entries of a read
// end up in pileup order (by locus), as a stable sort leaves them
__global__ void k_id_rank(Raw in, const uint32_t *dense, const uint32_t *id_off, const uint32_t *grouped,
                          const uint32_t *entry_locus, unsigned long long *skey, uint32_t *sval, uint32_t *sloc,
                          Scalars *sc) {
    const bool void_run = sc->id_exceeded != 0;
    for (uint32_t p = blockIdx.x * TPB + threadIdx.x; p < in.n_entries; p += gridDim.x * TPB) {
        if (void_run) {  // keep what follows inside its arrays until the host sees the flag and starts over
            skey[p] = p;
            sval[p] = p;
            sloc[p] = 0;
            continue;
        }
        const uint32_t e = grouped[p];
        const uint32_t d = dense[e];
        const uint32_t b = id_off[d], n = id_off[d + 1] - b;
        uint32_t rank = p - b;
        if (n > kRankScanLimit) {
            sc->regroup = 1;
        } else if (n > 1) {
            rank = 0;
            for (uint32_t q = b; q < b + n; ++q) rank += grouped[q] < e ? 1u : 0u;
        }
        skey[b + rank] = d;  // equal keys <=> same read; ascending in (chromosome, read id)
        sval[b + rank] = e;
        sloc[b + rank] = sloc_pack(entry_locus[e], in.id_base(e));  // the one gather by entry index
    }
}

// ---- stage 1 of the single-entry fast path: which ids repeat, and the reads of those that do ------------------
// Only WHICH ids occur more than once is wanted of all entries; how often and where only of the few that do. So the
// entries do not count their id and they need no atomic: every entry STORES its number in its id's slot (k_id_store:
// one of the entries of an id wins, whichever), then every entry looks whether the slot holds its own number
// (k_id_check): if not, its id occurs again -- it flags itself and the winner, so every entry of a repeated id ends up
// flagged, and the winner stands for the read. Two passes of plain coalesced stores and loads (the ids of
// neighbouring entries are neighbours) where one atomic per entry took 185-220 us on C3 (device-scope atomics are
// executed at the memory side), and the 48 MB table needs no zeroing: a slot is read only by entries that wrote it.
// The flags are a byte per entry IN PILEUP ORDER: numbering the M entries is a pass over 13 MB (k_flag_count,
// k_flag_scan, k_compact_m) where a scan that gathers hist[dense[e]] took 155 us; the reads of the M entries are
// the groups of equal winner (k_m_fields counts them per winner, a scan over the M entries, k_group_scatter,
// k_group_rank): no offsets over the id space (104 us), no histogram of the repeated ids.
constexpr uint32_t kFlagBlock = 4096;      // entries per workgroup of the numbering passes
// (a thread takes four consecutive entries, 16 bytes per lane and stream; `vec`: the caller's read-id array is 16-byte
// aligned -- the tables of this file are)
__global__ void k_id_store(Raw in, const uint32_t *entry_locus, const uint32_t *id_base, const uint32_t *id_negmin,
                           const Scalars *sc, uint32_t *dense, uint32_t *last, int vec) {
    if (sc->id_exceeded) return;  // the table is too small: no flags, no M entries; the caller starts over
    __shared__ ChrTables tables;
    const bool in_lds = chr_tables_load(in, id_base, id_negmin, tables);
    const uint32_t n = in.n_entries, stride = gridDim.x * TPB;
    for (uint32_t q = blockIdx.x * TPB + threadIdx.x; q < (n + 3u) / 4u; q += stride) {
        const uint32_t e0 = q * 4u;
        uint32_t l[4], id[4], d[4];
        if (vec && e0 + 4u <= n) {
            const uint4 lv = reinterpret_cast<const uint4 *>(entry_locus)[q];
            const uint4 iv = reinterpret_cast<const uint4 *>(in.read_ids)[q];
            l[0] = lv.x, l[1] = lv.y, l[2] = lv.z, l[3] = lv.w;
            id[0] = iv.x, id[1] = iv.y, id[2] = iv.z, id[3] = iv.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t e = min(e0 + (uint32_t)u, n - 1u);
                l[u] = entry_locus[e];
                id[u] = in.read_ids[e];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = dense_id(in, id_base, id_negmin, tables, in_lds, l[u], id[u]);
        if (e0 + 4u <= n) {
            reinterpret_cast<uint4 *>(dense)[q] = make_uint4(d[0], d[1], d[2], d[3]);
#pragma unroll
            for (int u = 0; u < 4; ++u) last[d[u]] = e0 + (uint32_t)u + 1u;
        } else {
            for (uint32_t u = 0; e0 + u < n; ++u) {
                dense[e0 + u] = d[u];
                last[d[u]] = e0 + u + 1u;
            }
        }
    }
}
__global__ void k_id_check(uint32_t n, const Scalars *sc, const uint32_t *dense, const uint32_t *last, uint8_t *multi) {
    if (sc->id_exceeded) return;
    const uint32_t stride = gridDim.x * TPB;
    for (uint32_t q = blockIdx.x * TPB + threadIdx.x; q < (n + 3u) / 4u; q += stride) {
        const uint32_t e0 = q * 4u;
        uint32_t d[4], w[4];
        if (e0 + 4u <= n) {
            const uint4 dv = reinterpret_cast<const uint4 *>(dense)[q];
            d[0] = dv.x, d[1] = dv.y, d[2] = dv.z, d[3] = dv.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = dense[min(e0 + (uint32_t)u, n - 1u)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = last[d[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = e0 + (uint32_t)u;
            if (e < n && w[u] != e + 1u) {
                multi[e] = 1;
                multi[w[u] - 1u] = 1;
            }
        }
    }
}
// flags per block of kFlagBlock entries (the flag array is zero-padded to whole blocks)
__global__ __launch_bounds__(TPB) void k_flag_count(const uint8_t *multi, uint32_t *block_sum) {
    __shared__ uint32_t part[TPB / 64];
    static_assert(kFlagBlock == TPB * 16, "a thread takes 16 consecutive flags");
    const uint4 w = reinterpret_cast<const uint4 *>(multi + (size_t)blockIdx.x * kFlagBlock)[threadIdx.x];
    uint32_t n = ((w.x * 0x01010101u) >> 24) + ((w.y * 0x01010101u) >> 24) + ((w.z * 0x01010101u) >> 24)
            + ((w.w * 0x01010101u) >> 24);
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0u) {
        for (int w2 = 1; w2 < TPB / 64; ++w2) n += part[w2];
        block_sum[blockIdx.x] = n;
    }
}
// exclusive offsets of the blocks (one workgroup; 3000 blocks on C3), the number of M entries
constexpr int TPB_SCAN = 1024;
__global__ __launch_bounds__(TPB_SCAN) void k_flag_scan(const uint32_t *block_sum, uint32_t n_blocks, uint32_t *block_off,
                                                       uint32_t *m_idx_end, Scalars *sc) {
    constexpr int TPB = TPB_SCAN;  // (this kernel's own)
    __shared__ uint32_t part[TPB / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0u) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_blocks; base += TPB) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sum[i] : 0u;
        uint32_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += up;
        }
        if (lane == 63u) part[wv] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w2 = 0; w2 < wv; ++w2) before += part[w2];
        if (i < n_blocks) block_off[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == TPB - 1) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0u) {
        *m_idx_end = carry;  // m_idx[E]
        sc->n_multi_id = carry;
    }
}
// m_idx = exclusive scan of the flags (entry e is an M entry iff m_idx[e + 1] != m_idx[e]) and the list of the M
// entries; k_m_fields, once their number is known, copies them out as a compacted pileup of their own and finds
// which of them holds its id's slot (the head its read is walked from)
__global__ __launch_bounds__(TPB) void k_compact_m(Raw in, const uint8_t *multi, const uint32_t *block_off,
                                                  uint32_t *m_idx, uint32_t *m_entry) {
    __shared__ uint32_t part[TPB / 64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t e0 = blockIdx.x * kFlagBlock + threadIdx.x * 16u;
    const uint4 w = reinterpret_cast<const uint4 *>(multi + (size_t)blockIdx.x * kFlagBlock)[threadIdx.x];
    const uint32_t words[4] = {w.x, w.y, w.z, w.w};
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) mine += (words[i] * 0x01010101u) >> 24;
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += up;
    }
    if (lane == 63u) part[wv] = incl;
    __syncthreads();
    uint32_t j = block_off[blockIdx.x] + incl - mine;
    for (uint32_t w2 = 0; w2 < wv; ++w2) j += part[w2];
    const uint32_t n = in.n_entries;
    // m_idx leaves through LDS: a thread's 16 consecutive words, stored from registers, were 16 stores of 64 lanes
    // 64 bytes apart (250 us on C3); rows of 17 words keep the transposing writes off one bank
    __shared__ uint32_t tile[TPB * 17];
    {
        uint32_t jj = j;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            tile[threadIdx.x * 17u + (uint32_t)i] = jj;
            jj += (words[i >> 2] >> ((i & 3) * 8)) & 1u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t i = (uint32_t)k * TPB + threadIdx.x;  // position in the block
        const uint32_t e = blockIdx.x * kFlagBlock + i;
        if (e < n) m_idx[e] = tile[(i >> 4) * 17u + (i & 15u)];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t e = e0 + (uint32_t)i;
        if (e >= n) break;  // (the padding holds no flags)
        if ((words[i >> 2] >> ((i & 3) * 8)) & 1u) m_entry[j++] = e;
    }
}
__global__ void k_m_fields(Raw in, const uint32_t *eloc, const uint32_t *dense, const uint32_t *last,
                           const uint32_t *m_idx, const uint32_t *m_entry, uint32_t n_m, uint32_t *rid_m,
                           uint32_t *idb_m, uint32_t *eloc_m, uint32_t *winner_m, uint32_t *count_m) {
    for (uint32_t j = blockIdx.x * TPB + threadIdx.x; j < n_m; j += gridDim.x * TPB) {
        const uint32_t e = m_entry[j];
        rid_m[j] = in.read_ids[e];
        idb_m[j] = in.id_base(e);
        eloc_m[j] = eloc[e];
        const uint32_t w = m_idx[last[dense[e]] - 1u];  // the M index of the entry that stands for this entry's id
        winner_m[j] = w;
        atomicAdd(&count_m[w], 1u);
    }
}
// the entries of a read next to each other (goff: exclusive sums of the counts; the counts are taken down again)
__global__ void k_group_scatter(const uint32_t *winner_m, uint32_t n_m, const uint32_t *goff, uint32_t *count_m,
                                uint32_t *members, uint32_t *g_begin, uint32_t *g_len) {
    for (uint32_t j = blockIdx.x * TPB + threadIdx.x; j < n_m; j += gridDim.x * TPB) {
        const uint32_t w = winner_m[j];
        const uint32_t b = goff[w], n = goff[w + 1] - b;
        members[b + atomicSub(&count_m[w], 1u) - 1u] = j;
        g_begin[j] = b;
        g_len[j] = n;
    }
}
// pileup order inside a read (the M indices grow with the pileup): a member's rank is the number of smaller members.
// skey: the read's first position stands for its id (equal keys <=> same read is all the stages behind need).
// (A read beyond kRankScanLimit keeps the order of the scatter and raises Scalars::regroup -- the attempt is void,
// the caller sorts --, but every position holds an entry.)
__global__ void k_group_rank(Raw sub, const uint32_t *members, const uint32_t *g_begin, const uint32_t *g_len,
                             const uint32_t *eloc_m, uint32_t n_m, unsigned long long *skey, uint32_t *sval,
                             uint32_t *sloc, Scalars *sc) {
    for (uint32_t p = blockIdx.x * TPB + threadIdx.x; p < n_m; p += gridDim.x * TPB) {
        const uint32_t m = members[p];
        const uint32_t b = g_begin[m], n = g_len[m];
        uint32_t rank = p - b;
        if (n > kRankScanLimit) {
            sc->regroup = 1;
        } else {
            rank = 0;
            for (uint32_t q = b; q < b + n; ++q) rank += members[q] < m ? 1u : 0u;
        }
        skey[b + rank] = b;
        sval[b + rank] = m;
        sloc[b + rank] = sloc_pack(eloc_m[m], sub.id_base(m));
    }
}

constexpr uint32_t kCibBits = 7;
constexpr uint32_t kNoEntry = 0xFFFFFFFFu;  // low word of an entry_kc slot: the entry was dropped
// Single-entry fast path: an S entry (the only entry of its read) needs no k -- nothing of its read is looked
// up again -- so the low word of its entry_kc slot carries kSingle alone and its base sits above
// (block, cell in block) in the high word; k_bin_place turns the rank into the tail flag.
constexpr uint32_t kSingle = 0x80000000u;
constexpr uint32_t kSingleBaseShift = 29;  // in the high word of entry_kc (block << 7 | cell in block below)
constexpr uint32_t kSingleTail = 4u;       // in the low word of a grouped S entry: kSingle | tail | base

// ---- the single-entry fast path ---------------------------------------------------------------------
// With sparse loci nine reads in ten have ONE entry (k_id_check says which), and for those the whole
// read assembly -- scatter, rank, duplicate rule, per-read lists, per-read info -- is the identity. Only the
// entries of ids that occur more than once ("M entries") go through it, as a compacted pileup of their own
// (the kernels below see a Raw whose entry arrays are the compacted copies); the others ("S entries") meet
// them again at the appearance-rank scan (every S entry is the first and only entry of its read) and at
// the binning.
// Appearance rank of a pileup entry = marked entries before it. Single-entry fast path: every S entry is
// marked, so rank(e) = e - (unmarked M entries before e) = e - unm[m_idx[e]], with unm the exclusive prefix of
// (1 - mark) over the M entries only -- a scan over 5 % of the pileup (C3) instead of one over all of it.
// Otherwise the ranks are an array (the scan over the marks).
struct Ranks {
    const uint32_t *m_idx, *unm;  // fast path (unm != null)
    const uint32_t *arank;        // or the array
    __device__ __forceinline__ uint32_t operator()(uint32_t e) const { return unm ? e - unm[m_idx[e]] : arank[e]; }
};
struct UnmarkedM {  // input of the scan that gives unm
    const uint32_t *mark_m;
    uint32_t n_m;
    __device__ __forceinline__ uint32_t operator()(uint32_t j) const { return j < n_m ? 1u - mark_m[j] : 0u; }
};
// appearance ranks of the M entries (k_read_info reads the rank of a read's first entry), the number of reads
__global__ void k_arank_m(const uint32_t *unm, const uint32_t *m_entry, uint32_t n_m, uint32_t n, uint32_t *arank_m,
                          Scalars *sc) {
    for (uint32_t j = blockIdx.x * TPB + threadIdx.x; j < n_m; j += gridDim.x * TPB) arank_m[j] = m_entry[j] - unm[j];
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->reads_total = n - unm[n_m];
}
// per chromosome the first appearance rank
__global__ void k_rbeg(Raw in, Ranks rank, uint32_t *rbeg) {
    for (uint32_t c = blockIdx.x * TPB + threadIdx.x; c <= in.n_chr; c += gridDim.x * TPB)
        rbeg[c] = rank((uint32_t)in.locus_entry_off[in.chr_locus_off[c]]);
}
// Input of the one scan over the sorted order: low word = first entry of a read (head of a run of
// equal keys), high word = the entry survives the duplicate rule. The inclusive sums give, per
// position, the 1-based read number and the number of kept entries up to and including it.
struct HeadKeepOp {
    const unsigned long long *skey;
    const uint32_t *keep;
    const uint32_t *split;  // 1 where a flush re-opens the id as a new read (k_split_update); null: nowhere
    __device__ __forceinline__ unsigned long long operator()(uint32_t s) const {
        const unsigned long long head = (s == 0 || skey[s] != skey[s - 1] || (split && split[s])) ? 1ull : 0ull;
        return head | ((unsigned long long)keep[s] << 32);
    }
};
__device__ __forceinline__ uint32_t incl_reads(unsigned long long v) { return (uint32_t)v; }
__device__ __forceinline__ uint32_t incl_kept(unsigned long long v) { return (uint32_t)(v >> 32); }

// duplicate-position rule (:387-395) per (read, locus) group of the sorted order, and the mark of
// every read's first entry (in pileup order) for the appearance rank
__global__ void k_dup_mark(Raw in, const unsigned long long *skey, const uint32_t *sval,
                           const uint32_t *sloc, const uint32_t *split, uint32_t n, uint32_t *keep,
                           uint32_t *mark) {
    for (uint32_t s = blockIdx.x * TPB + threadIdx.x; s < n; s += gridDim.x * TPB) {
        const uint32_t e = sval[s];
        const uint32_t l = sloc_locus(sloc[s]);
        const bool same_read = s > 0 && skey[s] == skey[s - 1];  // same id: a split starts at a new locus
        mark[e] = (same_read && !(split && split[s])) ? 0u : 1u;
        if (s == 0) mark[n] = 0u;
        if (same_read && sloc_locus(sloc[s - 1]) == l) continue;  // not a group head
        uint32_t stored = s;  // position of the stored entry of this (read, locus)
        bool have = true;
        keep[s] = 0u;
        for (uint32_t t = s + 1; t < n && skey[t] == skey[s] && sloc_locus(sloc[t]) == l; ++t) {
            keep[t] = 0u;
            if (have) {
                // second mate at the stored position: equal base -> ignored; different -> both go
                if (sloc_base(sloc[t]) != sloc_base(sloc[stored])) have = false;
            } else {
                stored = t;  // the position is free again: this entry is appended
                have = true;
            }
        }
        if (have) keep[stored] = 1u;
    }
}

// run starts (reads) and the per-read lists of kept entries (CSR payload)
__global__ void k_runs_csr(Raw in, const unsigned long long *incl, const uint32_t *sval,
                           const uint32_t *sloc, uint32_t n, uint32_t *run_start, uint32_t *read_locus,
                           uint8_t *read_base) {
    for (uint32_t s = blockIdx.x * TPB + threadIdx.x; s < n; s += gridDim.x * TPB) {
        const unsigned long long cur = incl[s], prev = s ? incl[s - 1] : 0ull;
        if (incl_reads(cur) != incl_reads(prev)) run_start[incl_reads(cur) - 1] = s;
        if (s == n - 1) run_start[incl_reads(cur)] = n;
        if (incl_kept(cur) != incl_kept(prev)) {
            const uint32_t k = incl_kept(prev), v = sloc[s];
            read_locus[k] = sloc_locus(v);
            read_base[k] = (uint8_t)sloc_base(v);
        }
    }
}

// per read: span check, offsets into the per-read lists, appearance rank, start position in rank
// order, entries of multi-locus reads; per chromosome the first rank
__global__ __launch_bounds__(TPB) void k_read_info(Raw in, const unsigned long long *incl, const uint32_t *run_start,
                                                  const uint32_t *sval, const uint32_t *sloc,
                                                  const uint32_t *arank, uint32_t mfl, uint32_t *run_rank,
                                                  uint32_t *rbeg, uint32_t *read_off, Scalars *sc) {
    __shared__ unsigned long long part[TPB / 64];
    __shared__ uint32_t part_len[TPB / 64];
    const uint32_t n = in.n_entries;
    const uint32_t n_runs = n ? incl_reads(incl[n - 1]) : 0u;
    unsigned long long multi = 0;
    uint32_t longest = 0;
    for (uint32_t r = blockIdx.x * TPB + threadIdx.x; r < n_runs; r += gridDim.x * TPB) {
        const uint32_t s0 = run_start[r], s1 = run_start[r + 1];
        const uint32_t e0 = sval[s0];
        const uint32_t p0 = in.locus_pos[sloc_locus(sloc[s0])], p1 = in.locus_pos[sloc_locus(sloc[s1 - 1])];
        // a read whose entries reach start + mfl can be flushed before its last entry arrives and is
        // then re-opened as a new read (:368-371, :379-382): k_split_update finds where
        if (p1 - p0 >= mfl) sc->long_reads = 1;
        if ((unsigned long long)p1 + mfl > 0xFFFFFFFFull) sc->need_host = 1;
        const uint32_t rk = arank[e0];
        run_rank[r] = rk;
        const uint32_t k0 = s0 ? incl_kept(incl[s0 - 1]) : 0u, k1 = incl_kept(incl[s1 - 1]);
        read_off[r] = k0;
        if (r == n_runs - 1) read_off[n_runs] = k1;
        if (k1 - k0 > 1) multi += k1 - k0;
        longest = max(longest, k1 - k0);
    }
    if (rbeg) {  // (the single-entry fast path has the ranks of all entries elsewhere: k_rbeg)
        for (uint32_t c = blockIdx.x * TPB + threadIdx.x; c <= in.n_chr; c += gridDim.x * TPB)
            rbeg[c] = arank[(uint32_t)in.locus_entry_off[in.chr_locus_off[c]]];
    }
    // one atomic per workgroup: same-address atomics are slow
    for (int off = 32; off > 0; off >>= 1) {
        multi += __shfl_down(multi, off);
        longest = max(longest, (uint32_t)__shfl_down(longest, off));
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = multi;
        part_len[threadIdx.x >> 6] = longest;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        uint32_t len = 0;
        for (int w = 0; w < TPB / 64; ++w) {
            sum += part[w];
            len = max(len, part_len[w]);
        }
        if (sum) atomicAdd(&sc->multi_entries, sum);
        if (len > 1) atomicMax(&sc->max_read_entries, len);  // reads of one entry are the rule: no atomic for them
    }
}

// completed-prefix count per locus (:348-352): reads of the chromosome with start + mfl <= position. A read
// starts at the position of its first entry, and `arank` counts the first entries in pileup order, so the
// reads that started at or before a locus are the ranks up to the end of that locus: a search over the
// chromosome's positions instead of a table of start positions per read.
__global__ void k_completed(Raw in, Ranks rank, const uint32_t *rbeg, uint32_t mfl, uint32_t *cnt, Scalars *sc) {
    for (uint32_t l = blockIdx.x * TPB + threadIdx.x; l < in.n_loci; l += gridDim.x * TPB) {
        const uint32_t c = last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, l);
        const uint32_t pos = in.locus_pos[l], l0 = in.chr_locus_off[c];
        if ((unsigned long long)pos + mfl > 0xFFFFFFFFull && in.locus_entry_off[l + 1] > in.locus_entry_off[l])
            sc->need_host = 1;  // start + mfl leaves 32 bits (the reference computes in uint32 there)
        uint32_t done = 0;
        if (pos >= mfl && in.locus_pos[l0] <= pos - mfl) {
            const uint32_t lp = l0 + last_le<uint32_t>(in.locus_pos + l0, l - l0 + 1, pos - mfl);
            done = rank((uint32_t)in.locus_entry_off[lp + 1]) - rbeg[c];
        }
        cnt[l] = done;
    }
}

// flush chain (:356-373): flushed = c(l) whenever c(l) - flushed >= 4 * num_threads. One workgroup
// per chromosome; the chain is sequential, so lane 0 walks LDS tiles of the counts.
// The loci at which a flush happens are listed per chromosome (flush_loci[l0 ..), flush_count[c]).
// (only when some id spans >= max_fragment_length -- Scalars::long_reads, set by k_read_info just
// before: nothing reads the list otherwise, and the bookkeeping triples the cost of the serial walk).
// Last, the first TAIL entry of the chromosome among the single-entry reads (tail_begin != null): such an entry
// is its read's first, the appearance ranks grow with the entry index, so "never flushed" -- rank - first rank of
// the chromosome >= flushed -- holds from one entry of the chromosome on. k_bin_place compares entry indices with
// it instead of looking every S entry's rank up. (A TPB-ary search: three rounds over 13 M entries.)
__global__ __launch_bounds__(TPB) void k_flush_chain(Raw in, const uint32_t *cnt, uint32_t threshold,
                                                    const Scalars *sc, uint32_t *flushed_out,
                                                    uint32_t *flush_loci, uint32_t *flush_count, Ranks rank,
                                                    const uint32_t *rbeg, uint32_t *tail_begin) {
    __shared__ uint32_t buf[2048];
    __shared__ uint32_t hits[2048];  // flush loci of the tile: the serial walk touches LDS only
    __shared__ uint32_t s_flushed, s_listed, s_hits;
    const uint32_t c = blockIdx.x;
    const uint32_t l0 = in.chr_locus_off[c], l1 = in.chr_locus_off[c + 1];
    if (threadIdx.x == 0) {
        s_flushed = 0;
        s_listed = 0;
    }
    for (uint32_t base = l0; base < l1; base += 2048) {
        const uint32_t n = min(2048u, l1 - base);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += TPB) buf[i] = cnt[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t f = s_flushed, h = 0;
            if (sc->long_reads) {
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t v = buf[i];
                    if (v - f >= threshold) {  // v >= f: the counts never decrease
                        f = v;
                        hits[h++] = base + i;
                    }
                }
            } else {
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t v = buf[i];
                    if (v - f >= threshold) f = v;
                }
            }
            s_flushed = f;
            s_hits = h;
        }
        __syncthreads();
        const uint32_t h = s_hits, listed = s_listed;
        for (uint32_t i = threadIdx.x; i < h; i += TPB) flush_loci[l0 + listed + i] = hits[i];
        __syncthreads();
        if (threadIdx.x == 0) s_listed = listed + h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        flushed_out[c] = s_flushed;
        flush_count[c] = s_listed;
    }
    if (tail_begin) {
        __shared__ uint32_t s_first;
        const unsigned long long want = (unsigned long long)rbeg[c] + s_flushed;  // the first rank that is never flushed
        uint32_t lo = (uint32_t)in.locus_entry_off[l0], hi = (uint32_t)in.locus_entry_off[l1];
        // invariant: entries before lo have a smaller rank, the entry hi (or the chromosome's end) has not
        while (lo < hi) {  // (uniform)
            const uint32_t step = (hi - lo + TPB - 1) / TPB;
            const unsigned long long probe = (unsigned long long)lo + (unsigned long long)threadIdx.x * step;
            const bool reached = probe < hi && (unsigned long long)rank((uint32_t)probe) >= want;
            __syncthreads();
            if (threadIdx.x == 0) s_first = TPB;
            __syncthreads();
            if (reached) atomicMin(&s_first, threadIdx.x);
            __syncthreads();
            const uint32_t first = s_first;
            if (first == TPB) {  // none of the probes: behind the last one
                lo = lo + ((hi - lo - 1) / step) * step + 1;
            } else {
                hi = lo + first * step;
                if (first) lo = lo + (first - 1) * step + 1;
            }
        }
        if (threadIdx.x == 0) tail_begin[c] = lo;
    }
}

// Where do flushes cut the reads whose entries span >= max_fragment_length? A flush at locus f (before
// its entries are added, :356-373) erases every live read with start + mfl <= position(f); an entry
// of an erased id opens a new read that starts there (:379-382). One thread per id walks the id's
// entries (a handful) against the chromosome's flush loci. The cuts change the read starts, hence the
// completed counts, hence possibly later flushes: the caller iterates until nothing changes.
__global__ void k_split_update(Raw in, const unsigned long long *skey, const uint32_t *sloc, uint32_t mfl, const uint32_t *flush_loci,
                               const uint32_t *flush_count, uint32_t *split, Scalars *sc) {
    const uint32_t n = in.n_entries;
    for (uint32_t s = blockIdx.x * TPB + threadIdx.x; s < n; s += gridDim.x * TPB) {
        if (s > 0 && skey[s] == skey[s - 1]) continue;  // not the first entry of an id
        uint32_t end = s + 1;
        while (end < n && skey[end] == skey[s]) ++end;
        if (end == s + 1) continue;
        const uint32_t l_first = sloc_locus(sloc[s]);
        const uint32_t p_first = in.locus_pos[l_first], p_last = in.locus_pos[sloc_locus(sloc[end - 1])];
        if (p_last - p_first < mfl) continue;  // never in a flushed prefix before its last entry
        const uint32_t c = last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, l_first);
        const uint32_t *fl = flush_loci + in.chr_locus_off[c];
        const uint32_t n_fl = flush_count[c];
        uint32_t seg_start = p_first, l_prev = l_first;
        for (uint32_t t = s + 1; t < end; ++t) {
            const uint32_t l_t = sloc_locus(sloc[t]);
            uint32_t cut = 0;
            if (l_t != l_prev) {
                // the first flush locus after l_prev whose position has reached seg_start + mfl
                uint32_t lo = 0, hi = n_fl;
                const unsigned long long reach = (unsigned long long)seg_start + mfl;
                while (lo < hi) {
                    const uint32_t mid = lo + (hi - lo) / 2;
                    if (fl[mid] <= l_prev || in.locus_pos[fl[mid]] < reach) lo = mid + 1; else hi = mid;
                }
                if (lo < n_fl && fl[lo] <= l_t) {
                    cut = 1;
                    seg_start = in.locus_pos[l_t];
                }
            }
            if (split[t] != cut) {
                split[t] = cut;
                sc->split_changed = 1;
            }
            l_prev = l_t;
        }
    }
}

// The binning key: cell block | locus (lbits bits) | cell in block (7 bits) -- bit fields, so that
// taking it apart costs shifts instead of 64-bit divisions.
__device__ __forceinline__ unsigned long long bin_key(uint32_t blk, uint32_t l, uint32_t cib, uint32_t lbits) {
    return ((((unsigned long long)blk << lbits) | l) << kCibBits) | cib;
}

// binning key (cell block, locus, cell) of every kept entry, in per-read (CSR) order; validates the
// group -> row mapping
// ... and, per kept entry, what k_records needs of its read in 5 bytes (appearance rank; base | multi-locus
// flag): the reads are adjacent here, in k_records' binned order they are five scattered gathers
__global__ void k_keys2(Raw in, const uint32_t *sval, const unsigned long long *incl,
                        const uint32_t *read_locus, const uint32_t *run_rank, const uint32_t *read_off,
                        uint32_t num_cells, uint32_t B,
                        uint32_t lbits, unsigned long long *key2, uint32_t *val2, uint32_t *t_read,
                        uint32_t *krank, uint8_t *kflags, unsigned long long *entry_kc, const uint32_t *entry_map,
                        uint32_t *blk_cnt_add, uint32_t L, Scalars *sc) {
    // entry_map: the pileup entry of a (compacted) entry of `in` (single-entry fast path), or null: itself
    const uint32_t n = in.n_entries;
    for (uint32_t s = blockIdx.x * TPB + threadIdx.x; s < n; s += gridDim.x * TPB) {
        const unsigned long long cur = incl[s], prev = s ? incl[s - 1] : 0ull;
        const uint32_t e_out = entry_map ? entry_map[sval[s]] : sval[s];
        if (incl_kept(cur) == incl_kept(prev)) {
            if (entry_kc) entry_kc[e_out] = kNoEntry;
            continue;
        }
        const uint32_t k = incl_kept(prev);
        const uint32_t group = in.id_base(sval[s]) >> 2;
        uint32_t cell = 0;
        if (group >= in.n_groups) {
            sc->error = 1;
        } else {
            cell = in.g2p[group];
            if (cell >= num_cells) {
                sc->error = 2;
                cell = 0;
            }
        }
        const uint32_t blk = cell / B, cib = cell - blk * B;
        if (key2) {  // the radix path sorts these; the counting path bins through entry_kc
            key2[k] = bin_key(blk, read_locus[k], cib, lbits);
            val2[k] = k;
        }
        const uint32_t r = incl_reads(cur) - 1;
        t_read[k] = r;
        krank[k] = run_rank[r];
        kflags[k] = (uint8_t)((in.id_base(sval[s]) & 3u) | (read_off[r + 1] - read_off[r] > 1u ? 4u : 0u));
        // counting path: back in pileup order, where the entries of a locus are adjacent; one 8-byte
        // scatter per entry carries k and (block, cell in block)
        if (entry_kc) entry_kc[e_out] = ((unsigned long long)((blk << kCibBits) | cib) << 32) | k;
        // (single-entry fast path: k_bin_hist has counted the S entries, possibly long ago; the few kept M
        // entries join with one scattered atomic each)
        if (blk_cnt_add) atomicAdd(&blk_cnt_add[(size_t)blk * (L + 1) + read_locus[k]], 1u);
    }
}

// counting path, kept entries by (cell block, locus). The entries of a locus are adjacent in the
// pileup, so a wave counts a locus' entries per cell block in LDS -- global atomics would all hit the
// handful of addresses of the loci in flight.
// Single-entry fast path (m_idx != null): ONLY the S entries are counted, and their slots are made here -- (block,
// cell) from the group map, validated as k_keys2 does for the M entries -- with nothing of the read assembly:
// k_keys2 adds the kept M entries afterwards, and k_bin_place knows an S entry's tail flag from its index.
// (Measured and dropped: this pass on a stream of its own beside stages 1b-4. The kernels there are not idle
// time but memory-side work -- the id-space scan went from 104 to 169 us next to it -- and the step gained
// nothing.)
// A workgroup takes TL consecutive loci at a time (TL a power of two, <= 64, nb * (TL + 1) words of LDS) and counts
// their entries -- one contiguous stretch of the pileup -- into hist[block][locus in tile], a thread per entry,
// four entries of a thread in flight; the locus of an entry comes from the tile's offsets in LDS (a search over 65
// words: a third of the pass' vector instructions, but reading k_entry_locus' table instead was SLOWER, 193 against
// 167 us -- the pass is bound by the memory streams it keeps open, see below). The tile leaves as rows of TL consecutive loci per block -- coalesced, where a column per locus touched nb
// different cache lines for 4 bytes each (6.3 M scattered stores on C3). Rows are TL + 1 words apart: the entries of
// one locus go to many blocks and a stride of TL would put them all on one bank.
// (Until round 3 a WAVE walked a locus at a time, sixteen loci one after the other with three dependent round trips
// each: 180 us on C3. What the pass costs is the sum of its memory streams, measured by switching them off one at a
// time in a twin launch into dead buffers: the 8-byte slot store 52 us, the group -> cell gather 37, the 2-byte
// id_base read 21, a 4-byte entry -> locus read 11, the LDS atomics and the count rows nothing, the empty loop 50.)
__device__ __forceinline__ uint32_t tile_locus(const uint32_t *s_off, uint32_t lo, uint32_t hi, uint32_t e) {
    // last j in [lo, hi) with s_off[j] <= e (s_off[lo] <= e; empty loci share an offset with their successor)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(TPB) void k_bin_hist(Raw in, uint32_t nb, uint32_t TL, unsigned long long *__restrict__ entry_kc,
                                                 const uint32_t *__restrict__ m_idx, uint32_t num_cells,
                                                 uint32_t B_log2, uint32_t *__restrict__ blk_cnt, Scalars *sc,
                                                 uint32_t g2p_lds) {
    extern __shared__ uint32_t lds_hist[];  // nb * (TL + 1); then (g2p_lds) the group -> cell map as 16-bit words
    __shared__ uint32_t s_off[65];          // first entry of each of the tile's loci, and the end
    static_assert(TPB >= 65, "one thread per offset");
    const uint32_t L = in.n_loci, TLP = TL + 1u, TL_log2 = 31u - (uint32_t)__clz((int)TL);
    const uint32_t n_tiles = (L + TL - 1u) / TL;
    constexpr int U = 4;
    // the group -> cell map from LDS (a workgroup then takes several tiles): the gather from global memory was 37
    // of this pass' 170 us on C3
    uint16_t *s_g2p = reinterpret_cast<uint16_t *>(lds_hist + nb * TLP);
    if (g2p_lds) {
        for (uint32_t i = threadIdx.x; i < in.n_groups; i += TPB) s_g2p[i] = (uint16_t)min(in.g2p[i], 0xFFFFu);
    }
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t l0 = tile * TL, n_l = min(TL, L - l0);
        for (uint32_t i = threadIdx.x; i < nb * TLP; i += TPB) lds_hist[i] = 0;
        if (threadIdx.x <= n_l) s_off[threadIdx.x] = (uint32_t)in.locus_entry_off[l0 + threadIdx.x];
        __syncthreads();
        const uint32_t eb = s_off[0], ee = s_off[n_l];
        for (uint32_t base = eb + threadIdx.x; base < ee; base += TPB * U) {
            uint32_t slot[U];  // block of the entry, or none
            constexpr uint32_t kNone = 0xFFFFFFFFu;
            if (m_idx) {
                uint32_t m0[U], m1[U], ib[U], cell[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t e = base + (uint32_t)u * TPB;
                    const bool in_tile = e < ee;
                    m0[u] = in_tile ? m_idx[e] : 0u;
                    m1[u] = in_tile ? m_idx[e + 1] : 1u;  // (outside the tile: as an M entry, skipped)
                    ib[u] = in_tile ? in.id_base(e) : 0u;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t group = ib[u] >> 2;
                    cell[u] = 0;
                    if (m1[u] == m0[u]) {
                        if (group >= in.n_groups) sc->error = 1;
                        else cell[u] = g2p_lds ? (uint32_t)s_g2p[group] : in.g2p[group];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    slot[u] = kNone;
                    if (m1[u] != m0[u]) continue;  // an M entry: k_keys2
                    const uint32_t e = base + (uint32_t)u * TPB;
                    uint32_t c = cell[u];
                    if (c >= num_cells) {
                        sc->error = 2;
                        c = 0;
                    }
                    const uint32_t blk = c >> B_log2, cib = c - (blk << B_log2);
                    entry_kc[e] = ((unsigned long long)(((ib[u] & 3u) << kSingleBaseShift) | (blk << kCibBits) | cib) << 32)
                            | kSingle;
                    slot[u] = blk;
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t e = base + (uint32_t)u * TPB;
                    const unsigned long long kc = e < ee ? entry_kc[e] : (unsigned long long)kNoEntry;
                    slot[u] = (uint32_t)kc != kNoEntry ? (uint32_t)(kc >> 32) >> kCibBits : kNone;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (slot[u] == kNone) continue;
                const uint32_t j = tile_locus(s_off, 0u, n_l, base + (uint32_t)u * TPB);
                atomicAdd(&lds_hist[slot[u] * TLP + j], 1u);
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb * TL; i += TPB) {
            const uint32_t b = i >> TL_log2, j = i & (TL - 1u);
            if (j < n_l) {
                blk_cnt[(size_t)b * (L + 1) + l0 + j] = lds_hist[b * TLP + j];
                if (l0 + j == L - 1) blk_cnt[(size_t)b * (L + 1) + L] = 0;  // the closing slot of the block's row
            }
        }
        __syncthreads();
    }
}

// ... and places them: position = group offset + LDS cursor (arbitrary order inside the group); the same
// tiles of loci, the offsets loaded as rows. grouped[pos] = (cell in block | locus << 7) << 32 | k
// It also finds the loci at which some cell has more than one kept entry (a bitmap over the cells per wave,
// bm_words words each; 0: no room, every locus is flagged): only there does k_entry_records have to count the
// entries of the same cell for the pair bound -- on sparse loci that scan of every group was 40 % of its time.
// The bitmap wants a locus at a time, so a wave takes a run of consecutive loci of the tile -- one contiguous stretch
// of entries -- in chunks of 64 with the chunks of the next round already asked for (D in flight), and inside a chunk
// goes through the loci it holds in order; a locus is closed (flag written, bitmap cleared) when the chunk that
// holds its last entry is done.
__global__ __launch_bounds__(TPB) void k_bin_place(Raw in, uint32_t nb, uint32_t TL,
                                                  const unsigned long long *__restrict__ entry_kc,
                                                  const uint32_t *__restrict__ blk_off, uint32_t B_log2, uint32_t bm_words,
                                                  const uint32_t *tail_begin,
                                                  unsigned long long *__restrict__ grouped, uint8_t *dupflag) {
    extern __shared__ uint32_t lds_hist[];  // cursors, nb * (TL + 1); then the waves' cell bitmaps
    __shared__ uint32_t s_off[65];          // first entry of each of the tile's loci, and the end
    __shared__ uint32_t s_tail[64];         // per locus of the tile: the first tail entry of its chromosome (S entries)
    constexpr uint32_t NW = TPB / 64;
    constexpr int D = 4;  // chunks of a wave in flight
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, L = in.n_loci, TLP = TL + 1u;
    const uint32_t TL_log2 = 31u - (uint32_t)__clz((int)TL);
    uint32_t *bm = lds_hist + nb * TLP + wv * bm_words;
    for (uint32_t i = lane; i < bm_words; i += 64u) bm[i] = 0;
    const uint32_t n_tiles = (L + TL - 1u) / TL;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t l0 = tile * TL, n_l = min(TL, L - l0);
        for (uint32_t i = threadIdx.x; i < nb * TL; i += TPB) {
            const uint32_t b = i >> TL_log2, j = i & (TL - 1u);
            if (j < n_l) lds_hist[b * TLP + j] = blk_off[(size_t)b * (L + 1) + l0 + j];
        }
        if (threadIdx.x <= n_l) s_off[threadIdx.x] = (uint32_t)in.locus_entry_off[l0 + threadIdx.x];
        if (threadIdx.x < n_l)
            s_tail[threadIdx.x] = tail_begin
                    ? tail_begin[last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, l0 + threadIdx.x)] : 0u;
        __syncthreads();
        const uint32_t per_wave = (n_l + NW - 1u) / NW;
        const uint32_t ja = min(n_l, wv * per_wave), jb = min(n_l, ja + per_wave);
        const uint32_t E0 = s_off[ja], E1 = s_off[jb];
        uint32_t cur = ja;                // the locus being filled (wave-uniform)
        bool twice = bm_words == 0u;      // ... has some cell twice (per lane until the locus is closed)
        auto close_locus = [&]() {  // every entry of locus `cur` has been placed
            const bool any_twice = __ballot(twice) != 0ull;
            if (lane == 0u) dupflag[l0 + cur] = any_twice ? 1 : 0;
            if (bm_words) {  // the next locus starts from an empty bitmap
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = lane; i < bm_words; i += 64u) bm[i] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            twice = bm_words == 0u;
            ++cur;
        };
        auto chunk = [&](unsigned long long kc, uint32_t base) {  // the 64 entries from `base`, loaded
            const uint32_t e = base + lane;
            const bool valid = e < E1;
            const uint32_t j = valid ? tile_locus(s_off, cur, jb, e) : 0xFFFFFFFFu;
            const uint32_t chunk_end = min(E1, base + 64u);
            while (true) {  // (wave-uniform)
                if (valid && j == cur) {
                    uint32_t k = (uint32_t)kc;
                    if (k != kNoEntry) {
                        uint32_t cc = (uint32_t)(kc >> 32);
                        if (k & kSingle) {  // entry index -> tail flag (k_flush_chain); base beside it
                            k = kSingle | (e >= s_tail[cur] ? kSingleTail : 0u) | (cc >> kSingleBaseShift);
                            cc &= (1u << kSingleBaseShift) - 1u;
                        }
                        if (bm_words) {
                            const uint32_t cell = ((cc >> kCibBits) << B_log2) + (cc & ((1u << kCibBits) - 1u));
                            const uint32_t bit = 1u << (cell & 31u);
                            twice |= (atomicOr(&bm[cell >> 5], bit) & bit) != 0u;
                        }
                        const uint32_t pos = atomicAdd(&lds_hist[(cc >> kCibBits) * TLP + cur], 1u);
                        // cell in block | locus << 7 above k (k_entry_records; the radix path never comes here)
                        grouped[pos] = ((unsigned long long)((cc & ((1u << kCibBits) - 1u)) | ((l0 + cur) << kCibBits)) << 32) | k;
                    }
                }
                if (s_off[cur + 1] > chunk_end) break;  // the locus goes on in the next chunk
                close_locus();
                if (cur >= jb) break;
            }
        };
        unsigned long long ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t e = E0 + (uint32_t)d * 64u + lane;
            ring[d] = e < E1 ? entry_kc[e] : (unsigned long long)kNoEntry;
        }
        for (uint32_t base = E0; base < E1; base += 64u * D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const uint32_t b = base + (uint32_t)d * 64u;
                if (b >= E1) break;  // (wave-uniform)
                const unsigned long long kc = ring[d];
                const uint32_t nxt = b + 64u * D + lane;
                ring[d] = nxt < E1 ? entry_kc[nxt] : (unsigned long long)kNoEntry;
                chunk(kc, b);
            }
        }
        while (cur < jb) close_locus();  // (a run without entries)
        __syncthreads();
    }
}

// After the binning sort: entries per (block, locus) group -> blk_cnt (no atomics: the head of a
// group counts its run), and sum over loci of (entries of one cell at the locus)^2 per cell (the
// Cauchy-Schwarz pair bound), accumulated in LDS per workgroup: a workgroup's slice of the sorted
// order spans only a few cell blocks.
__global__ __launch_bounds__(TPB) void k_group_counts(const unsigned long long *skey2, uint32_t n, uint32_t B,
                                                     uint32_t L, uint32_t lbits, uint32_t *blk_cnt,
                                                     unsigned long long *per_cell_sq) {
    constexpr uint32_t SLOTS = 512;
    __shared__ unsigned long long sq[SLOTS];
    __shared__ uint32_t first_cell;
    const uint32_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t d0 = blockIdx.x * per_block, d1 = min(n, d0 + per_block);
    for (uint32_t i = threadIdx.x; i < SLOTS; i += TPB) sq[i] = 0;
    if (threadIdx.x == 0 && d0 < n) {
        const unsigned long long key = skey2[d0];
        first_cell = (uint32_t)(key >> (kCibBits + lbits)) * B;  // first cell of the slice's first block
    }
    __syncthreads();
    for (uint32_t d = d0 + threadIdx.x; d < d1; d += TPB) {
        const unsigned long long key = skey2[d];
        const unsigned long long grp = key >> kCibBits;  // (block, locus)
        const uint32_t blk = (uint32_t)(grp >> lbits), l = (uint32_t)grp & ((1u << lbits) - 1u);
        if (d == 0 || skey2[d - 1] >> kCibBits != grp) {
            uint32_t len = 1;
            for (uint32_t t = d + 1; t < n && skey2[t] >> kCibBits == grp; ++t) ++len;
            blk_cnt[(size_t)blk * (L + 1) + l] = len;
        }
        if (d == 0 || skey2[d - 1] != key) {
            unsigned long long cnt = 1;
            for (uint32_t t = d + 1; t < n && skey2[t] == key; ++t) ++cnt;
            const uint32_t cell = blk * B + ((uint32_t)key & ((1u << kCibBits) - 1u));
            const uint32_t rel = cell - first_cell;
            if (rel < SLOTS) atomicAdd(&sq[rel], cnt * cnt);
            else atomicAdd(&per_cell_sq[cell], cnt * cnt);
        }
    }
    __syncthreads();
    if (d0 < n) {
        for (uint32_t i = threadIdx.x; i < SLOTS; i += TPB) {
            if (sq[i]) atomicAdd(&per_cell_sq[first_cell + i], sq[i]);
        }
    }
}

// Locus ranges (one partition shared by all cell blocks). The loci are cut into segments of
// cap_loci loci, one workgroup per segment; inside its segment a workgroup cuts greedily: the
// longest range from `s` in which no block has more than cap_entries entries (feasibility is
// monotone in the range end; the search gallops from the previous range's length); a locus that
// exceeds the cap alone becomes a single-locus range. k_ranges_compact concatenates the segments.
//
// The staging limits depend on the tile variant, and the tile variant on the pair bound, which is
// still on the device: the kernels pick the limits themselves (CapChoice), the host learns the
// outcome with the final read-back.
struct CapChoice {
    uint32_t entries_plain, loci_plain;    // int64 tile (or the masks variant)
    uint32_t entries_counts, loci_counts;  // count tile
    unsigned long long count_limit;        // count tile iff allow_counts and pair bound < count_limit
    uint32_t allow_counts;
    __device__ __forceinline__ bool counts(const Scalars *sc) const {
        return allow_counts && sc->pair_bound < count_limit;
    }
};

// The pair bound that decides between the two sets of limits comes out of the grouping that is still
// running when the offsets are ready, so the ranges are cut for BOTH sets (blockIdx.y: 0 plain, 1 count
// tile) on the side stream meanwhile, and k_ranges_compact picks one.
__global__ __launch_bounds__(TPB) void k_ranges_segment(const uint32_t *blk_off, uint32_t nb, uint32_t L,
                                                       CapChoice caps, uint32_t *seg_ends_both,
                                                       uint32_t *seg_count_both, size_t variant_stride) {
    const bool use_counts = blockIdx.y == 1u;
    const uint32_t cap_entries = use_counts ? caps.entries_counts : caps.entries_plain;
    const uint32_t cap_loci = use_counts ? caps.loci_counts : caps.loci_plain;
    uint32_t *seg_ends = seg_ends_both + blockIdx.y * variant_stride;
    uint32_t *seg_count = seg_count_both + blockIdx.y * variant_stride;
    const size_t stride = (size_t)L + 1;
    const uint32_t seg = blockIdx.x;
    if ((unsigned long long)seg * cap_loci >= L) {  // the grid is sized for the smaller of the two limits
        if (threadIdx.x == 0) seg_count[seg] = 0;
        return;
    }
    const uint32_t seg_begin = seg * cap_loci, seg_end = min(L, seg_begin + cap_loci);
    uint32_t *ends = seg_ends + (size_t)seg * cap_loci;  // at most cap_loci ranges per segment
    uint32_t s = seg_begin, nr = 0, guess = 0;
    auto feasible = [&](uint32_t e) {
        int ok = 1;
        for (uint32_t b = threadIdx.x; b < nb; b += TPB)
            if (blk_off[b * stride + e] - blk_off[b * stride + s] > cap_entries) ok = 0;
        return __syncthreads_and(ok) != 0;
    };
    while (s < seg_end) {
        uint32_t e = s + 1;
        if (feasible(e)) {
            const uint32_t top = seg_end;
            uint32_t lo = e, hi = top;  // lo feasible, answer in [lo, hi]
            if (feasible(top)) {
                lo = top;
            } else if (guess > 1 && s + guess < top) {  // bracket around the previous length
                hi = top - 1;
                const uint32_t g = s + guess;
                if (feasible(g)) {
                    lo = g;
                    uint32_t step = max(guess / 8, 1u);
                    while (lo < hi) {  // gallop up
                        const uint32_t nx = min(hi, lo + step);
                        if (feasible(nx)) lo = nx; else { hi = nx - 1; break; }
                        step *= 2;
                    }
                } else {
                    hi = g - 1;
                    uint32_t step = max(guess / 8, 1u);
                    while (hi > lo) {  // gallop down
                        const uint32_t nx = (hi - lo > step) ? hi - step : lo;
                        if (feasible(nx)) { lo = nx; break; }
                        hi = nx - 1;
                        step *= 2;
                    }
                }
            } else {
                hi = top - 1;
            }
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo + 1) / 2;
                if (feasible(mid)) lo = mid; else hi = mid - 1;
            }
            e = lo;
        }
        guess = e - s;
        if (threadIdx.x == 0) ends[nr] = e;
        ++nr;
        s = e;
    }
    if (threadIdx.x == 0) seg_count[seg] = nr;
}

__global__ __launch_bounds__(TPB) void k_ranges_compact(const uint32_t *seg_ends_both, const uint32_t *seg_count_both,
                                                       size_t variant_stride, uint32_t n_seg, CapChoice caps,
                                                       uint32_t *range_off, const unsigned long long *per_cell_sq,
                                                       uint32_t n_cells_padded, int force_variant, Scalars *sc) {
    // force_variant < 0: the pair bound first (the largest per-cell sum of squares), then the limits it allows;
    // 0 / 1: the limits of the int64 tile / of the count tile, whatever the bound (the counting path cuts for
    // the count tile before the bound is known, k_pair_bound checks afterwards)
    __shared__ unsigned long long s_best[TPB / 64];
    if (force_variant < 0) {
        unsigned long long best = 0;
        for (uint32_t i = threadIdx.x; i < n_cells_padded; i += TPB) best = max(best, per_cell_sq[i]);
        for (int off = 32; off > 0; off >>= 1) best = max(best, (unsigned long long)__shfl_down(best, off));
        if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < TPB / 64; ++w) best = max(best, s_best[w]);
            sc->pair_bound = max(sc->pair_bound, best);
        }
        __syncthreads();
    }
    const bool use_counts = force_variant < 0 ? caps.counts(sc) : force_variant == 1;
    const uint32_t cap_loci = use_counts ? caps.loci_counts : caps.loci_plain;
    const uint32_t *seg_ends = seg_ends_both + (use_counts ? variant_stride : 0);
    const uint32_t *seg_count = seg_count_both + (use_counts ? variant_stride : 0);
    __shared__ uint32_t s_base;
    if (threadIdx.x == 0) {
        s_base = 0;
        range_off[0] = 0;
    }
    __syncthreads();
    for (uint32_t seg = 0; seg < n_seg; ++seg) {
        const uint32_t base = s_base, cnt = seg_count[seg];
        for (uint32_t i = threadIdx.x; i < cnt; i += TPB) range_off[1 + base + i] = seg_ends[(size_t)seg * cap_loci + i];
        __syncthreads();
        if (threadIdx.x == 0) s_base = base + cnt;
        __syncthreads();
    }
    if (threadIdx.x == 0) sc->num_ranges = s_base;
}

// per locus: its chromosome and its index inside its locus range (k_records looks both up per entry)
// (locus_rel: the range-relative locus in the low half; bit 31: some cell has two kept entries at the locus --
// k_bin_place's flag, so that k_entry_records reads ONE table per entry)
constexpr uint32_t kLocusDup = 0x80000000u;
__global__ void k_locus_info(Raw in, const uint32_t *range_off, const Scalars *sc, const uint8_t *dupflag,
                             uint32_t *locus_chr, uint32_t *locus_rel) {
    const uint32_t n_ranges = sc->num_ranges;
    for (uint32_t l = blockIdx.x * TPB + threadIdx.x; l < in.n_loci; l += gridDim.x * TPB) {
        locus_chr[l] = last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, l);
        locus_rel[l] = (l - range_off[last_le<uint32_t>(range_off, n_ranges + 1, l)])
                | ((dupflag && dupflag[l]) ? kLocusDup : 0u);
    }
}

// entry records at their final (binned) position d: window masks from the per-read lists
__global__ void k_records(Raw in, const unsigned long long *skey2, const uint32_t *sval2, uint32_t n,
                          const uint32_t *t_read, const uint32_t *read_off, const uint32_t *read_locus,
                          const uint8_t *read_base, const uint32_t *krank, const uint8_t *kflags, const uint32_t *rbeg,
                          const uint32_t *flushed, const uint32_t *locus_chr, const uint32_t *locus_rel, uint32_t B,
                          uint32_t lbits, uint4 *entry, uint32_t *entry32, uint32_t *mask32,
                          uint32_t *entry_read, Scalars *sc) {
    uint32_t n_wide = 0;
    for (uint32_t d = blockIdx.x * TPB + threadIdx.x; d < n; d += gridDim.x * TPB) {
        const unsigned long long key = skey2[d];
        const uint32_t k = sval2[d];
        const uint32_t cib = (uint32_t)key & ((1u << kCibBits) - 1u);
        const unsigned long long grp = key >> kCibBits;
        const uint32_t blk = (uint32_t)(grp >> lbits), l = (uint32_t)grp & ((1u << lbits) - 1u);
        const uint32_t cell = blk * B + cib;
        const uint32_t fl = kflags[k];
        const uint32_t base = fl & 3u;
        const bool multi = (fl & 4u) != 0u;
        const uint32_t chr = locus_chr[l];
        const bool tail = krank[k] - rbeg[chr] >= flushed[chr];
        // the read's list of kept entries: only a multi-locus read has neighbours to look for
        uint32_t r = 0, lo = k, hi = k + 1;
        if (multi) {
            r = t_read[k];
            lo = read_off[r];
            hi = read_off[r + 1];
        }
        uint32_t meta = cell | (base << kMetaBaseShift) | (tail ? kMetaTail : 0u);
        uint32_t masks = 0, bases = 0;
        bool wide = false;
        for (uint32_t j = k; j-- > lo;) {
            const uint32_t dist = l - read_locus[j];
            if (dist > kNarrowWindow) wide = true;
            if (dist > kWindow) {
                meta |= kMetaPrevOvf;
                break;
            }
            masks |= 1u << (dist - 1);
        }
        for (uint32_t j = k + 1; j < hi; ++j) {
            const uint32_t dist = read_locus[j] - l;
            if (dist > kNarrowWindow) wide = true;
            if (dist > kWindow) {
                meta |= kMetaNextOvf;
                break;
            }
            masks |= 1u << (16 + dist - 1);
            bases |= (uint32_t)(read_base[j] & 1u) << (dist - 1);
            bases |= (uint32_t)((read_base[j] >> 1) & 1u) << (16 + dist - 1);
        }
        // the 16-byte record and the read index serve the flagged entries only: pairs of two multi-locus
        // reads, and (correct_tiles) pairs of two reads that were never flushed
        if (multi || tail) {
            entry[d] = make_uint4(meta, masks, bases, l);
            entry_read[d] = r;  // (read only for pairs of two multi-locus reads)
        }
        entry32[d] = cib | (base << kC_BaseShift) | (tail ? kC_Tail : 0u) | (multi ? kC_Multi : 0u)
                | (wide ? kC_Wide : 0u) | ((locus_rel[l] & 0xFFFFu) << 16);
        if (mask32)  // staged by the clustered-loci tile variant only
            mask32[d] = (masks & 0xFFu) | (((masks >> 16) & 0xFFu) << 8) | ((bases & 0xFFu) << 16)
                    | (((bases >> 16) & 0xFFu) << 24);
        n_wide += wide ? 1u : 0u;
    }
    if (n_wide) atomicAdd(&sc->n_wide, n_wide);
}

// ---- counting path: k_bin_rank + k_records in one pass ------------------------------------------------
// Nothing depends on the order of the entries inside a (cell block, locus) group (the pair kernels enumerate
// every pair of a locus once, in whatever order the entries stand; diagonal tiles pair an entry with the
// entries after it), so the entries stay where k_bin_place put them: no ranking, no sorted copy of keys and
// values. k_bin_place packs the locus next to (cell in block, k) in its 8 bytes, the block follows from the
// position (a search over the nb block boundaries, held in LDS), and a thread per entry scans its group for
// the entries of the same cell: an entry of a cell with n entries at the locus adds n to the cell's sum, n of
// them n^2 -- the Cauchy-Schwarz pair bound. (A thread per GROUP instead read the tables once per group but
// walked its entries one after the other, a chain of gathers per thread: 7.4 ms on C3 against 2.7.)
// What the record of a kept M entry k needs of its read, made where the read's entries are adjacent (the per-read
// lists, k ascending) instead of in binned order, where it is a chain of five dependent gathers per entry -- a
// twentieth of the entries on sparse loci, but every wave of the record pass holds a few and waited for them
// (90 us of its 280 on C3). m_rec[k] = {base, tail, overflow flags as in an entry's meta word | kRecMulti | kRecWide,
// window masks, window bases, read index}; k_entry_records adds the cell and the locus.
constexpr uint32_t kRecMulti = 1u << 30, kRecWide = 1u << 31;
__global__ void k_m_records(Raw in, uint32_t n_kept_m, const uint32_t *t_read, const uint32_t *read_off,
                            const uint32_t *read_locus, const uint8_t *read_base, const uint32_t *krank,
                            const uint8_t *kflags, const uint32_t *rbeg, const uint32_t *flushed, uint4 *m_rec,
                            Scalars *sc) {
    // the chromosomes' first loci from LDS (up to 1024 of them): searched in global memory the chromosome of an entry
    // was a chain of dependent loads in front of everything else the thread does (C3 clustered: 68 M entries)
    __shared__ uint32_t s_chr[1025];
    const bool chr_lds = in.n_chr <= 1024u;
    if (chr_lds)
        for (uint32_t i = threadIdx.x; i <= in.n_chr; i += TPB) s_chr[i] = in.chr_locus_off[i];
    __syncthreads();
    uint32_t n_wide = 0;  // (rare: an atomic per thread that met one; the count rides on the packing's last read-back)
    for (uint32_t k = blockIdx.x * TPB + threadIdx.x; k < n_kept_m; k += gridDim.x * TPB) {
        const uint32_t fl = kflags[k], l = read_locus[k];
        const bool multi = (fl & 4u) != 0u;
        const uint32_t chr = chr_lds ? last_le<uint32_t>(s_chr, in.n_chr + 1, l)
                                     : last_le<uint32_t>(in.chr_locus_off, in.n_chr + 1, l);
        const bool tail = krank[k] - rbeg[chr] >= flushed[chr];
        // the read's list of kept entries: only a multi-locus read has neighbours to look for
        uint32_t r = 0, lo = k, hi = k + 1;
        if (multi) {
            r = t_read[k];
            lo = read_off[r];
            hi = read_off[r + 1];
        }
        uint32_t meta = ((fl & 3u) << kMetaBaseShift) | (tail ? kMetaTail : 0u) | (multi ? kRecMulti : 0u);
        uint32_t masks = 0, bases = 0;
        for (uint32_t j = k; j-- > lo;) {
            const uint32_t dist = l - read_locus[j];
            if (dist > kNarrowWindow) meta |= kRecWide;
            if (dist > kWindow) {
                meta |= kMetaPrevOvf;
                break;
            }
            masks |= 1u << (dist - 1);
        }
        for (uint32_t j = k + 1; j < hi; ++j) {
            const uint32_t dist = read_locus[j] - l;
            if (dist > kNarrowWindow) meta |= kRecWide;
            if (dist > kWindow) {
                meta |= kMetaNextOvf;
                break;
            }
            masks |= 1u << (16 + dist - 1);
            bases |= (uint32_t)(read_base[j] & 1u) << (dist - 1);
            bases |= (uint32_t)((read_base[j] >> 1) & 1u) << (16 + dist - 1);
        }
        m_rec[k] = make_uint4(meta, masks, bases, r);
        n_wide += (meta & kRecWide) ? 1u : 0u;
    }
    if (n_wide) atomicAdd(&sc->n_wide, n_wide);
}

struct RecordTables {  // by value: what a record needs beside the group's own entries
    const uint4 *m_rec;  // per kept M entry (k_m_records)
    const uint32_t *locus_rel;
    uint4 *entry;
    uint32_t *entry32, *mask32, *entry_read;
};

// rec: the M entry's m_rec word group, loaded by the caller (all zero for an S entry)
__device__ __forceinline__ void emit_record(const RecordTables &t, uint32_t d, uint32_t k, uint4 rec, uint32_t cib,
                                            uint32_t cell, uint32_t l, uint32_t lrel) {
    uint32_t base, meta, masks = 0, bases = 0, r = 0;
    bool multi = false, wide = false, tail;
    if (k & kSingle) {  // single-entry fast path: all there is to know came along (k_bin_place)
        base = k & 3u;
        tail = (k & kSingleTail) != 0u;
        meta = cell | (base << kMetaBaseShift) | (tail ? kMetaTail : 0u);
    } else {
        base = (rec.x >> kMetaBaseShift) & 3u;
        tail = (rec.x & kMetaTail) != 0u;
        multi = (rec.x & kRecMulti) != 0u;
        wide = (rec.x & kRecWide) != 0u;
        meta = cell | (rec.x & ~(kRecMulti | kRecWide));
        masks = rec.y;
        bases = rec.z;
        r = rec.w;
    }
    // the 16-byte record and the read index serve the flagged entries only (see k_records)
    if (multi || tail) {
        t.entry[d] = make_uint4(meta, masks, bases, l);
        t.entry_read[d] = r;
    }
    t.entry32[d] = cib | (base << kC_BaseShift) | (tail ? kC_Tail : 0u) | (multi ? kC_Multi : 0u)
            | (wide ? kC_Wide : 0u) | (lrel << 16);
    if (t.mask32)
        t.mask32[d] = (masks & 0xFFu) | (((masks >> 16) & 0xFFu) << 8) | ((bases & 0xFFu) << 16)
                | (((bases >> 16) & 0xFFu) << 24);
}

constexpr int TPB_REC = 512;
__global__ __launch_bounds__(TPB_REC) void k_entry_records(const unsigned long long *grouped, const uint32_t *blk_off,
                                                          uint32_t n, uint32_t nb, uint32_t L, uint32_t B,
                                                          uint32_t lbits, RecordTables t,
                                                          unsigned long long *per_cell_sq, Scalars *sc) {
    constexpr uint32_t SLOTS = 512;
    __shared__ unsigned long long sq[SLOTS];
    __shared__ uint32_t blk_start[1025];  // first entry of every cell block (nb <= 1024), and the end
    __shared__ uint32_t first_cell;
    const uint32_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t d0 = blockIdx.x * per_block, d1 = min(n, d0 + per_block);
    for (uint32_t i = threadIdx.x; i < SLOTS; i += TPB_REC) sq[i] = 0;
    for (uint32_t i = threadIdx.x; i <= nb; i += TPB_REC) blk_start[i] = i < nb ? blk_off[(size_t)i * (L + 1)] : n;
    __syncthreads();
    auto block_of = [&](uint32_t p) {  // last block whose first entry is <= p (empty blocks share a start)
        uint32_t lo = 0, hi = nb;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (blk_start[mid] <= p) lo = mid; else hi = mid;
        }
        return lo;
    };
    if (threadIdx.x == 0 && d0 < n) first_cell = block_of(d0) * B;
    __syncthreads();
    const uint32_t lmask = (1u << lbits) - 1u;
    // A thread has U entries in flight, and an entry's chain of dependent reads is as short as it can be: its slot,
    // then ONE per-locus word (range-relative locus | duplicate flag) and, for an M entry (one in twenty on sparse
    // loci), the 16 bytes k_m_records made of its read; only at a locus with a duplicate cell are the group's bounds
    // read and its members counted. (Round 2 read group bounds, three per-locus tables and two per-chromosome words
    // for every entry and walked the read's lists here: 280 us on C3; switching the pieces off one at a time in a
    // twin launch showed 92 us for the per-locus level of the chain, 90 for the M entries' five levels, 126 for
    // the stores, 20 for the block search. Staging the slice's per-locus words in LDS -- the slots are in (block,
    // locus) order, a slice covers 2000 consecutive loci -- was measured SLOWER, 197 against 176 us: two more
    // dependent reads and two barriers in front of every slice.)
    constexpr int U = 4;
    for (uint32_t base = d0; base < d1; base += TPB_REC * U) {
        unsigned long long mine[U];
        uint32_t blk[U], linfo[U], same[U];
        uint4 rec[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = base + (uint32_t)u * TPB_REC + threadIdx.x;
            mine[u] = p < d1 ? grouped[p] : (unsigned long long)kSingle;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = base + (uint32_t)u * TPB_REC + threadIdx.x;
            const uint32_t l = (uint32_t)(mine[u] >> (32 + kCibBits)) & lmask;
            blk[u] = p < d1 ? block_of(p) : 0u;
            linfo[u] = t.locus_rel[l];
            rec[u] = make_uint4(0u, 0u, 0u, 0u);
            if (((uint32_t)mine[u] & kSingle) == 0u) rec[u] = t.m_rec[(uint32_t)mine[u]];  // an M entry
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = base + (uint32_t)u * TPB_REC + threadIdx.x;
            same[u] = 1;
            if (p < d1 && (linfo[u] & kLocusDup)) {  // some cell has two kept entries at this locus: count mine
                const uint32_t l = (uint32_t)(mine[u] >> (32 + kCibBits)) & lmask;
                const uint32_t cib = (uint32_t)(mine[u] >> 32) & ((1u << kCibBits) - 1u);
                const size_t g = (size_t)blk[u] * (L + 1) + l;
                const uint32_t gb = blk_off[g], ge = blk_off[g + 1];
                if (ge - gb > kRankScanLimit) {
                    sc->regroup = 1;
                } else if (ge - gb > 1) {
                    same[u] = 0;
                    for (uint32_t q = gb; q < ge; ++q)
                        same[u] += ((uint32_t)(grouped[q] >> 32) & ((1u << kCibBits) - 1u)) == cib ? 1u : 0u;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = base + (uint32_t)u * TPB_REC + threadIdx.x;
            if (p >= d1) continue;
            const uint32_t k = (uint32_t)mine[u], cib = (uint32_t)(mine[u] >> 32) & ((1u << kCibBits) - 1u);
            const uint32_t l = (uint32_t)(mine[u] >> (32 + kCibBits)) & lmask;
            const uint32_t cell = blk[u] * B + cib;
            emit_record(t, p, k, rec[u], cib, cell, l, linfo[u] & 0xFFFFu);
            const uint32_t rel = cell - first_cell;
            if (rel < SLOTS) atomicAdd(&sq[rel], (unsigned long long)same[u]);
            else atomicAdd(&per_cell_sq[cell], (unsigned long long)same[u]);
        }
    }
    __syncthreads();
    if (d0 < n)
        for (uint32_t i = threadIdx.x; i < SLOTS; i += TPB_REC)
            if (sq[i]) atomicAdd(&per_cell_sq[first_cell + i], sq[i]);
}

// the pair bound (largest per-cell sum of squares), and whether the locus ranges, cut for the count tile
// before it was known, have to be cut again
__global__ __launch_bounds__(TPB) void k_pair_bound(const unsigned long long *per_cell_sq, uint32_t n_cells_padded,
                                                   CapChoice caps, uint32_t assumed_counts, Scalars *sc) {
    __shared__ unsigned long long s_best[TPB / 64];
    unsigned long long best = 0;
    for (uint32_t i = threadIdx.x; i < n_cells_padded; i += TPB) best = max(best, per_cell_sq[i]);
    for (int off = 32; off > 0; off >>= 1) best = max(best, (unsigned long long)__shfl_down(best, off));
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) best = max(best, s_best[w]);
        sc->pair_bound = max(sc->pair_bound, best);
        sc->caps_wrong = (assumed_counts && !caps.counts(sc)) ? 1u : 0u;
    }
}

// after the ranges were cut again: the range-relative locus of every entry (bits 16-31 of the compact form)
__global__ __launch_bounds__(TPB) void k_fix_locus_rel(const uint32_t *blk_off, uint32_t nb, uint32_t L,
                                                      const uint32_t *locus_rel, uint32_t *entry32) {
    const uint32_t n_groups = nb * (L + 1);
    for (uint32_t g = blockIdx.x * TPB + threadIdx.x; g < n_groups; g += gridDim.x * TPB) {
        const uint32_t l = g % (L + 1);
        // (the closing slot of a block's row is no group -- and behind the last one blk_off ends: its successor
        // is whatever the allocation held, which a fresh handle showed as a memory fault in
        // test_very_deep_loci_unstaged_ranges)
        if (l == L) continue;
        const uint32_t b = blk_off[g], e = blk_off[g + 1];
        if (b == e) continue;
        const uint32_t lrel = (locus_rel[l] & 0xFFFFu) << 16;
        for (uint32_t d = b; d < e; ++d) entry32[d] = (entry32[d] & 0xFFFFu) | lrel;
    }
}

int bits_for(unsigned long long max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e__); \
    } while (0)

// One attempt at the pipeline. With force_radix == false the two groupings (entries by read,
// kept entries by (cell block, locus)) use the counting scheme when its preconditions hold; a group
// too long for it sets *retry and the caller runs the attempt again with the radix sorts.
std::string pack_attempt(const DeviceFlatPileup &in, uint32_t num_cells, uint32_t mfl, uint32_t num_threads,
                         uint32_t block_cells, StageGeometry (*geometry)(uint32_t), bool allow_count_tile,
                         bool force_radix, bool no_assumptions, hipStream_t stream, DevicePacked *out,
                         bool *need_host, int *retry) {
    *retry = kNoRetry;
    const uint32_t E = static_cast<uint32_t>(in.n_entries);
    const uint32_t L = in.n_loci, C = in.n_chr;
    DevicePacked &pk = *out;
    // worst-case block count (64-cell blocks) for buffers sized before the tile size is chosen
    const size_t n_off_max = (size_t)((num_cells + 63) / 64) * ((size_t)L + 1);
    enum { KEY_A, KEY_B, VAL_A, VAL_B, ELOC, WORK_A, WORK_B, RUNS, CUB, TMP, MISC, BIN, ENTRY_KC,
           M_IDX, M_ENTRY, RID_M, IDB_M, ELOC_M, DENSE_M, MARK_M, ARANK_M, DUPF, SEGS };
    auto &S = pk.scratch;
    // the counting scheme for read ids needs a table over the id space
    const size_t id_space_cap = std::min<size_t>((size_t)kIdSpaceFactor * E + 1024, (size_t)1 << 30);

    Raw raw{in.chr_locus_off, C, in.locus_pos, in.locus_entry_off, in.read_ids, in.id_base16,
            in.id_base32, in.group_id_to_pos, in.n_groups, L, E};

    // ---- buffers (sized up front: a re-allocation in mid-pipeline would synchronise) -----------
    // MISC: Scalars | id_max[C] | id_negmin[C] | id_base[C+1] | rbeg[C+1] | flushed[C] | cnt[L] |
    //       locus_chr[L] | locus_rel[L] | flush_loci[L] | flush_count[C] | tail_begin[C]
    HIP_OK(S[MISC].ensure(sizeof(Scalars) + sizeof(uint32_t) * ((size_t)7 * C + 4 + (size_t)4 * L) + 64));
    Scalars *sc = S[MISC].as<Scalars>();
    uint32_t *id_max = reinterpret_cast<uint32_t *>(sc + 1);
    uint32_t *id_negmin = id_max + C;
    uint32_t *id_base = id_negmin + C;
    uint32_t *rbeg = id_base + C + 1;
    uint32_t *flushed = rbeg + C + 1;
    uint32_t *cnt = flushed + C;
    uint32_t *locus_chr = cnt + L, *locus_rel = locus_chr + L;
    uint32_t *flush_loci = locus_rel + L, *flush_count = flush_loci + L;
    uint32_t *tail_begin = flush_count + C;
    // KEY_A: sort keys in, later mark[E+1] | arank[E+1], later the per-(block, locus) counts
    // (marks and appearance ranks, 2 (E + 1) words, with the (block, locus) counts behind them: the ranks are
    // still read when the counts are written, k_bin_hist)
    HIP_OK(S[KEY_A].ensure(std::max<size_t>((size_t)E * 8, ((size_t)2 * E + 4 + n_off_max + 1) * 4)));
    // KEY_B: sorted keys, later (counting path) the kept entries grouped by (block, locus)
    HIP_OK(S[KEY_B].ensure((size_t)E * 8));
    HIP_OK(S[VAL_A].ensure(std::max<size_t>(E, 64) * 4));
    HIP_OK(S[VAL_B].ensure((size_t)E * 4));
    // ELOC: entry -> locus (the pileup's entries: the binning passes read it to the end)
    HIP_OK(S[ELOC].ensure((size_t)E * 4 + 16));
    HIP_OK(S[WORK_A].ensure(((size_t)E + 1) * 4));
    HIP_OK(S[WORK_B].ensure(((size_t)2 * E + 2) * 4));
    // RUNS: run_start[R+1] | run_rank[R] | starts_by_rank[R], R <= E
    HIP_OK(S[RUNS].ensure(((size_t)3 * E + 8) * 4));
    // ENTRY_KC: locus per entry in sorted order (stages 2-4), later (k, cell) per entry (counting path)
    HIP_OK(S[ENTRY_KC].ensure((size_t)E * 8));
    uint32_t *sloc = S[ENTRY_KC].as<uint32_t>();
    // TMP: read index per kept entry; BIN: key2 x2, val2 x2, per-cell squares
    HIP_OK(S[TMP].ensure((size_t)E * 4 + 64));
    HIP_OK(S[BIN].ensure((size_t)E * 24 + ((size_t)num_cells + 130) * 8 + 64));
    unsigned long long *key_a = S[KEY_A].as<unsigned long long>(), *key_b = S[KEY_B].as<unsigned long long>();
    uint32_t *val_a = S[VAL_A].as<uint32_t>(), *val_b = S[VAL_B].as<uint32_t>();
    uint32_t *eloc = S[ELOC].as<uint32_t>();
    {
        size_t need = 0, most = 0;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, key_a, key_b, val_a, val_b, (int)E, 0, 64, stream));
        most = std::max(most, need);
        {
            hipcub::CountingInputIterator<uint32_t> positions(0u);
            hipcub::TransformInputIterator<unsigned long long, HeadKeepOp, hipcub::CountingInputIterator<uint32_t>>
                    flags(positions, HeadKeepOp{key_b, val_a, val_a});
            HIP_OK(hipcub::DeviceScan::InclusiveSum(nullptr, need, flags, key_a, (int)E, stream));
            most = std::max(most, need);
        }
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, val_a, val_b, (int)E + 1, stream));
        most = std::max(most, need);
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, val_a, val_b,
                                                (int)std::max<size_t>(n_off_max, id_space_cap + 2), stream));
        most = std::max(most, need);
        HIP_OK(S[CUB].ensure(most + 1024));
    }
    void *cub_tmp = S[CUB].p;
    size_t cub_cap = 0;
    HIP_OK(hipMemsetAsync(sc, 0, sizeof(Scalars) + sizeof(uint32_t) * ((size_t)3 * C + 1), stream));

    if (!pk.side) {
        HIP_OK(hipStreamCreateWithFlags(&pk.side, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&pk.ev_fork, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&pk.ev_join, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&pk.ev_offsets, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&pk.ev_flush, hipEventDisableTiming));
    }
    auto bin_tiles = [&](uint32_t nb_, uint32_t *TL_, uint32_t *grid_, size_t *lds_) {
        // tiles of TL consecutive loci per workgroup: nb * (TL + 1) words of LDS, 32 KiB at most (nb <= 1024)
        uint32_t TL = 64;
        while (TL > 1 && (size_t)nb_ * (TL + 1) * 4 > 32768) TL >>= 1;
        *TL_ = TL;
        *grid_ = std::min<uint32_t>((L + TL - 1) / TL, 8192);
        *lds_ = (size_t)nb_ * (TL + 1) * 4;
    };

    // ---- 1: entries grouped by (chromosome, read id), pileup order inside a read ---------------
    const HostTrace trace;
    trace.mark("begin");
    hipLaunchKernelGGL(k_entry_locus, dim3(blocks_for((uint64_t)L * 64)), dim3(TPB), 0, stream, raw, eloc, sc);
    hipLaunchKernelGGL(k_id_range, dim3(std::min<uint32_t>(256, (E + 4095) / 4096)), dim3(TPB), 0, stream, raw,
                       eloc, id_max, id_negmin, (reinterpret_cast<uintptr_t>(raw.read_ids) & 15u) == 0 ? 1 : 0);
    // The size of the id space decides between the counting scheme and the radix sort and sizes the
    // histogram. A handle that has packed before assumes the size of the previous call and does not wait
    // (k_id_bases raises Scalars::id_exceeded if that was too small: the attempt is then void and
    // repeated with the read-back).
    const bool assume = !force_radix && !no_assumptions && pk.id_space_hint && pk.id_space_hint <= id_space_cap;
    hipLaunchKernelGGL(k_id_bases, dim3(1), dim3(64), 0, stream, C, id_max, id_negmin, id_base,
                       (unsigned long long)(assume ? pk.id_space_hint : 0), sc);
    Scalars hsc;
    std::memset(&hsc, 0, sizeof(hsc));
    if (!assume) {
        HIP_OK(read_scalars(pk, stream, sc, nullptr, &hsc, nullptr));  // read-back 1: the size of the id space
        if (hsc.error == 3) return "positions must be strictly increasing within a chromosome";
    }
    const size_t id_space = assume ? (size_t)pk.id_space_hint
                                   : (size_t)std::min<unsigned long long>(hsc.id_space, 1ull << 40);
    const bool counting = !force_radix && id_space <= id_space_cap;
    HIP_OK(S[WORK_A].ensure(std::max<size_t>((size_t)E + 1, counting ? id_space + 1 : 0) * 4));
    HIP_OK(S[WORK_B].ensure(std::max<size_t>((size_t)2 * E + 2, counting ? id_space + 1 : 0) * 4));
    uint32_t *work_a = S[WORK_A].as<uint32_t>(), *work_b = S[WORK_B].as<uint32_t>();
    // The single-entry fast path (counting scheme only; SECEDO_PACK_SPLIT=0 turns it off): `sub` is the pileup
    // the read assembly of stages 1-4 sees -- the whole one, or the compacted entries of the ids that occur more
    // than once (n_m of them; m_entry maps them back, m_idx is the exclusive scan of their flags).
    static const bool split_allowed = [] {
        const char *e = std::getenv("SECEDO_PACK_SPLIT");
        return !(e && std::atoi(e) == 0);
    }();
    if (!pk.split_pays && ++pk.calls_without_split >= 16) {  // (pileups change: look again now and then)
        pk.split_pays = true;
        pk.calls_without_split = 0;
    }
    bool split_singles = counting && split_allowed && pk.split_pays;
    Raw sub = raw;
    const uint32_t *sub_eloc = eloc;
    uint32_t n_m = E;
    uint32_t *m_idx = nullptr, *m_entry = nullptr;
    if (counting) {
        uint32_t *hist = work_a, *id_off = work_b, *grouped = val_a;
        uint32_t *dense = S[KEY_A].as<uint32_t>();  // the radix path's unsorted keys live here
        if (!split_singles) HIP_OK(hipMemsetAsync(hist, 0, (id_space + 1) * 4, stream));
        if (split_singles) {
            // which ids repeat (k_id_store, k_id_check), the entries of those as a compacted pileup of their own
            const size_t padded = ((size_t)E + kFlagBlock - 1) / kFlagBlock * kFlagBlock + 16;
            const uint32_t n_flag_blocks = (uint32_t)((padded - 16) / kFlagBlock);
            HIP_OK(S[M_IDX].ensure(((size_t)E + 2) * 4));
            m_idx = S[M_IDX].as<uint32_t>();
            HIP_OK(S[MARK_M].ensure(std::max(((size_t)E + 2) * 4, padded)));
            uint8_t *multi = S[MARK_M].as<uint8_t>();  // (the marks of the M entries come later: k_dup_mark)
            HIP_OK(hipMemsetAsync(multi, 0, padded, stream));
            hipLaunchKernelGGL(k_id_store, dim3(blocks_for((E + 3) / 4)), dim3(TPB), 0, stream, raw, eloc, id_base, id_negmin,
                               sc, dense, hist, (reinterpret_cast<uintptr_t>(raw.read_ids) & 15u) == 0 ? 1 : 0);
            hipLaunchKernelGGL(k_id_check, dim3(blocks_for((E + 3) / 4)), dim3(TPB), 0, stream, E, sc, dense, hist, multi);
            for (int a : {M_ENTRY, RID_M, IDB_M, ELOC_M, ARANK_M}) HIP_OK(S[a].ensure((size_t)E * 4 + 16));
            // (the reads of the M entries, at most half of all entries:
            // count[n_m+1] | goff[n_m+1] | winner | begin | len | members)
            HIP_OK(S[DENSE_M].ensure(((size_t)6 * (E / 2 + 1) + 8) * 4));
            m_entry = S[M_ENTRY].as<uint32_t>();
            uint32_t *block_sum = grouped, *block_off = grouped + n_flag_blocks + 1;  // (VAL_A: 2 words per 4096 entries)
            hipLaunchKernelGGL(k_flag_count, dim3(n_flag_blocks), dim3(TPB), 0, stream, multi, block_sum);
            hipLaunchKernelGGL(k_flag_scan, dim3(1), dim3(TPB_SCAN), 0, stream, block_sum, n_flag_blocks, block_off, m_idx + E,
                               sc);
            hipLaunchKernelGGL(k_compact_m, dim3(n_flag_blocks), dim3(TPB), 0, stream, raw, multi, block_off, m_idx, m_entry);
            // read-back 1b: how many entries take the general path (sizes every launch over them)
            HIP_OK(read_scalars(pk, stream, sc, nullptr, &hsc, nullptr));
            if (hsc.error == 3) return "positions must be strictly increasing within a chromosome";
            if (hsc.id_exceeded) {
                pk.id_space_hint = 0;
                *retry = kRetrySameScheme;
                return std::string();
            }
            n_m = hsc.n_multi_id;
            if ((uint64_t)n_m * 2 > E) {  // clustered loci: (nearly) every read has several entries, no short cut
                split_singles = false;
                pk.split_pays = false;
                pk.calls_without_split = 0;
                n_m = E;
                m_idx = nullptr;
                m_entry = nullptr;
                // (the id slots hold entry numbers, the general path wants counts)
                HIP_OK(hipMemsetAsync(hist, 0, (id_space + 1) * 4, stream));
            }
        }
        if (split_singles) {
            sub = raw;
            sub.read_ids = S[RID_M].as<uint32_t>();
            sub.id_base16 = nullptr;
            sub.id_base32 = S[IDB_M].as<uint32_t>();
            sub.n_entries = n_m;
            sub_eloc = S[ELOC_M].as<uint32_t>();
            if (n_m) {
                // the reads of the M entries: the groups of equal winner
                uint32_t *count_m = S[DENSE_M].as<uint32_t>(), *goff = count_m + n_m + 1, *winner_m = goff + n_m + 1;
                uint32_t *g_begin = winner_m + n_m, *g_len = g_begin + n_m, *members = g_len + n_m;
                HIP_OK(hipMemsetAsync(count_m, 0, ((size_t)n_m + 1) * 4, stream));
                hipLaunchKernelGGL(k_m_fields, dim3(blocks_for(n_m)), dim3(TPB), 0, stream, raw, eloc, dense, hist, m_idx, m_entry,
                                   n_m, S[RID_M].as<uint32_t>(), S[IDB_M].as<uint32_t>(), S[ELOC_M].as<uint32_t>(), winner_m,
                                   count_m);
                cub_cap = S[CUB].bytes;
                HIP_OK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_cap, count_m, goff, (int)n_m + 1, stream));
                hipLaunchKernelGGL(k_group_scatter, dim3(blocks_for(n_m)), dim3(TPB), 0, stream, winner_m, n_m, goff, count_m,
                                   members, g_begin, g_len);
                hipLaunchKernelGGL(k_group_rank, dim3(blocks_for(n_m)), dim3(TPB), 0, stream, sub, members, g_begin, g_len,
                                   sub_eloc, n_m, key_b, val_b, sloc, sc);
            }
        } else {
            hipLaunchKernelGGL(k_id_hist, dim3(blocks_for(E)), dim3(TPB), 0, stream, raw, eloc, id_base, id_negmin, sc,
                               dense, hist);
            cub_cap = S[CUB].bytes;
            HIP_OK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_cap, hist, id_off, (int)(id_space + 1), stream));
            hipLaunchKernelGGL(k_id_scatter, dim3(blocks_for(E)), dim3(TPB), 0, stream, dense, E, id_off, sc, hist,
                               grouped);
            hipLaunchKernelGGL(k_id_rank, dim3(blocks_for(E)), dim3(TPB), 0, stream, raw, dense, id_off, grouped, eloc,
                               key_b, val_b, sloc, sc);
        }
    } else {
        const uint32_t id_bits = (uint32_t)bits_for(hsc.max_read_id);
        hipLaunchKernelGGL(k_entry_keys, dim3(blocks_for(E)), dim3(TPB), 0, stream, raw, eloc, id_bits, key_a, val_a);
        cub_cap = S[CUB].bytes;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_cap, key_a, key_b, val_a, val_b, (int)E, 0,
                                                   (int)id_bits + bits_for(C), stream));
        hipLaunchKernelGGL(k_sorted_locus, dim3(blocks_for(E)), dim3(TPB), 0, stream, raw, val_b, eloc, E, sloc);
    }
    const unsigned long long *skey = key_b;
    const uint32_t *sval = val_b;

    // ---- 2-4: duplicate rule, reads = runs of equal key, per-read lists, appearance ranks ------
    // (the number of reads R stays on the device until read-back 2; buffers indexed by read are laid
    // out for the upper bound E)
    uint32_t *keep = work_a;
    unsigned long long *incl = reinterpret_cast<unsigned long long *>(work_b);  // alive until k_keys2
    uint32_t *mark = S[KEY_A].as<uint32_t>();  // the unsorted keys / dense ids are dead
    uint32_t *arank = mark + (E + 1);
    uint32_t *run_start = S[RUNS].as<uint32_t>();
    uint32_t *run_rank = run_start + E + 1;
    // cuts of reads that outlive max_fragment_length: none on the first build (null), later the flags of
    // k_split_update in TMP, which is free until k_keys2 writes the read index per kept entry there
    uint32_t *split = nullptr;
    HIP_OK(pk.read_off.ensure(((size_t)E + 1) * 4));
    HIP_OK(pk.read_locus.ensure((size_t)E * 4));
    HIP_OK(pk.read_base.ensure(E));
    HIP_OK(pk.range_off.ensure(((size_t)L + 2) * 4));
    HIP_OK(pk.entry.ensure((size_t)E * 16));
    HIP_OK(pk.entry32.ensure((size_t)E * 4));
    HIP_OK(pk.mask32.ensure((size_t)E * 4));
    HIP_OK(pk.entry_read.ensure((size_t)E * 4));
    HIP_OK(pk.blk_off.ensure((n_off_max + 1) * 4));
    uint32_t *read_off = pk.read_off.as<uint32_t>(), *read_locus = pk.read_locus.as<uint32_t>();
    uint8_t *read_base = pk.read_base.as<uint8_t>();
    // reads from the current `split` flags, then completed counts and the flush chain. The chain is
    // sequential (one lane per chromosome) and only the final gather needs its result: it runs on a
    // side stream, next to the grouping of the kept entries.
    // (`sub` / n_m: the entries that go through the read assembly -- all of them, or with the single-entry
    // fast path those of the ids that occur more than once; marks and ranks are always those of the whole pileup)
    const Ranks ranks{split_singles ? m_idx : nullptr, split_singles ? arank : nullptr, arank};
    auto build_reads = [&]() -> std::string {
        uint32_t *mark_sub = split_singles ? S[MARK_M].as<uint32_t>() : mark;
        if (n_m) {
            hipLaunchKernelGGL(k_dup_mark, dim3(blocks_for(n_m)), dim3(TPB), 0, stream, sub, skey, sval, sloc, split, n_m,
                               keep, mark_sub);
            hipcub::CountingInputIterator<uint32_t> positions(0u);
            hipcub::TransformInputIterator<unsigned long long, HeadKeepOp, hipcub::CountingInputIterator<uint32_t>>
                    flags(positions, HeadKeepOp{skey, keep, split});
            cub_cap = S[CUB].bytes;
            HIP_OK(hipcub::DeviceScan::InclusiveSum(cub_tmp, cub_cap, flags, incl, (int)n_m, stream));
        }
        cub_cap = S[CUB].bytes;
        const uint32_t *arank_sub = arank;
        if (split_singles) {  // (the ranks' array holds the n_m + 1 prefix sums `unm` instead: struct Ranks)
            hipcub::CountingInputIterator<uint32_t> entries(0u);
            hipcub::TransformInputIterator<uint32_t, UnmarkedM, hipcub::CountingInputIterator<uint32_t>> unmarked(
                    entries, UnmarkedM{mark_sub, n_m});
            HIP_OK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_cap, unmarked, arank, (int)n_m + 1, stream));
            uint32_t *arank_m = S[ARANK_M].as<uint32_t>();
            hipLaunchKernelGGL(k_arank_m, dim3(blocks_for(std::max(n_m, 1u))), dim3(TPB), 0, stream, arank, m_entry, n_m,
                               E, arank_m, sc);
            arank_sub = arank_m;
            hipLaunchKernelGGL(k_rbeg, dim3(1), dim3(TPB), 0, stream, raw, ranks, rbeg);
        } else {
            HIP_OK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_cap, mark, arank, (int)E + 1, stream));
        }
        if (n_m) {
            hipLaunchKernelGGL(k_runs_csr, dim3(blocks_for(n_m)), dim3(TPB), 0, stream, sub, incl, sval, sloc, n_m,
                               run_start, read_locus, read_base);
            hipLaunchKernelGGL(k_read_info, dim3(std::min<uint32_t>(blocks_for(n_m), 2048)), dim3(TPB), 0, stream, sub,
                               incl, run_start, sval, sloc, arank_sub, mfl, run_rank, split_singles ? nullptr : rbeg,
                               read_off, sc);
        }
        HIP_OK(hipEventRecord(pk.ev_fork, stream));
        HIP_OK(hipStreamWaitEvent(pk.side, pk.ev_fork, 0));
        hipLaunchKernelGGL(k_completed, dim3(blocks_for(L)), dim3(TPB), 0, pk.side, raw, ranks, rbeg, mfl, cnt, sc);
        hipLaunchKernelGGL(k_flush_chain, dim3(C), dim3(TPB), 0, pk.side, raw, cnt, 4u * num_threads, sc, flushed,
                           flush_loci, flush_count, ranks, rbeg, split_singles ? tail_begin : nullptr);
        HIP_OK(hipEventRecord(pk.ev_flush, pk.side));
        HIP_OK(hipEventRecord(pk.ev_join, pk.side));
        return std::string();
    };
    {
        const std::string err = build_reads();
        if (!err.empty()) return err;
    }
    struct SideJoin {  // every way out of this function leaves the side stream idle
        hipStream_t side;
        bool joined = false;
        ~SideJoin() { if (!joined) (void)hipStreamSynchronize(side); }
    } side_join{pk.side};

    // read-back 2: status flags, R, number of kept entries, multi-locus statistics
    unsigned long long totals = 0;  // reads | kept entries << 32
    trace.mark("reads built (launched)");
    HIP_OK(read_scalars(pk, stream, sc, n_m ? incl + (n_m - 1) : nullptr, &hsc, &totals));
    trace.mark("read-back 2 arrived");
    if (hsc.error == 3) return "positions must be strictly increasing within a chromosome";
    if (hsc.id_exceeded) {
        pk.id_space_hint = 0;
        *retry = kRetrySameScheme;
        return std::string();
    }
    pk.id_space_hint = counting ? std::max<uint64_t>(hsc.id_space, 1) : 0;
    if (hsc.regroup) {
        *retry = kRetryRadix;
        return std::string();
    }
    if (hsc.need_host) {
        *need_host = true;
        return std::string();
    }
    if (hsc.long_reads) {
        // Some id's entries span >= max_fragment_length: a flush may erase it before its last entry and
        // the id re-opens as a new read. Cut at the flush loci, rebuild, repeat until the cuts are stable
        // (they move forward with the flushes they cause: a few rounds).
        bool stable = false;
        split = S[TMP].as<uint32_t>();
        HIP_OK(hipMemsetAsync(split, 0, (size_t)std::max(n_m, 1u) * 4, stream));
        for (int round = 0; round < kMaxSplitRounds && !stable; ++round) {
            HIP_OK(hipStreamWaitEvent(stream, pk.ev_join, 0));  // the chain of the previous build
            HIP_OK(hipMemsetAsync(&sc->split_changed, 0, 4, stream));
            hipLaunchKernelGGL(k_split_update, dim3(blocks_for(n_m)), dim3(TPB), 0, stream, sub, skey, sloc, mfl,
                               flush_loci, flush_count, split, sc);
            HIP_OK(read_scalars(pk, stream, sc, nullptr, &hsc, nullptr));
            if (!hsc.split_changed) {
                stable = true;
                break;
            }
            HIP_OK(hipMemsetAsync(&sc->multi_entries, 0, 8, stream));  // k_read_info adds to it
            const std::string err = build_reads();
            if (!err.empty()) return err;
            HIP_OK(read_scalars(pk, stream, sc, n_m ? incl + (n_m - 1) : nullptr, &hsc, &totals));
        }
        if (!stable) {
            *need_host = true;  // did not settle: the exact sequential emulation decides
            return std::string();
        }
    }
    // (single-entry fast path: `totals` counts the reads and kept entries of the M entries; every S entry is a
    // read and a kept entry of its own)
    const uint32_t kept_m = (uint32_t)(totals >> 32);
    const uint32_t R = split_singles ? hsc.reads_total : (uint32_t)totals;
    const uint32_t n_kept = split_singles ? kept_m + (E - n_m) : kept_m;
    const size_t nk = std::max<uint32_t>(n_kept, 1);
    pk.multi_entries = hsc.multi_entries;
    pk.max_read_entries = std::max(hsc.max_read_entries, n_kept ? 1u : 0u);
    if (block_cells == 0) {
        const StageGeometry g64 = geometry(64);
        const bool clustered = n_kept && (double)pk.multi_entries > g64.masks_threshold * (double)n_kept;
        block_cells = (clustered || num_cells <= 64) ? 64 : 128;
    }
    const StageGeometry geo = geometry(block_cells);
    const uint32_t B = block_cells, nb = (num_cells + B - 1) / B;
    if (B != 64 && B != 128) return "block_cells must be 64 or 128";
    const uint32_t B_log2 = B == 64 ? 6u : 7u;
    const uint32_t lbits = (uint32_t)bits_for(L - 1);
    // k_bin_place packs the locus into 25 bits at most, and bit 31 of a k marks an S entry
    if (!force_radix && (lbits + kCibBits > 32u || E >= kSingle)) {
        *retry = kRetryRadix;
        return std::string();
    }
    pk.num_cells = num_cells;
    pk.block_cells = B;
    pk.num_blocks = nb;
    pk.num_loci = L;
    pk.num_entries = n_kept;
    pk.num_reads = R;
    pk.stage_masks = n_kept && (double)pk.multi_entries > geo.masks_threshold * (double)n_kept;

    // ---- 5: kept entries grouped by (cell block, locus, cell), offsets, bound, ranges, records ----
    uint32_t *t_read = S[TMP].as<uint32_t>();
    unsigned long long *key2_a = S[BIN].as<unsigned long long>();
    unsigned long long *key2_b = key2_a + nk;
    uint4 *m_rec = S[BIN].as<uint4>();  // counting path (no binning keys): 16 bytes per kept M entry, k_m_records
    unsigned long long *per_cell_sq = key2_b + nk;  // nb * B entries
    uint32_t *val2_a = reinterpret_cast<uint32_t *>(per_cell_sq + (size_t)nb * B + 1);
    uint32_t *val2_b = val2_a + nk;
    const size_t n_off = (size_t)nb * ((size_t)L + 1);
    uint32_t *blk_off = pk.blk_off.as<uint32_t>();
    uint32_t *blk_cnt = S[KEY_A].as<uint32_t>() + 2 * ((size_t)E + 1);  // behind the marks and ranks
    // counting path: per pileup entry its k and cell (entry -> locus and the per-read scratch are dead)
    unsigned long long *entry_kc = force_radix ? nullptr : S[ENTRY_KC].as<unsigned long long>();
    HIP_OK(S[DUPF].ensure((size_t)L + 16));
    uint8_t *dupflag = S[DUPF].as<uint8_t>();  // per locus: some cell has two kept entries there (k_bin_place)
    // The pair bound decides the tile variant, and the tile variant the staging limits of the locus ranges;
    // the bound comes out of the grouping below, the ranges need only the offsets: they are cut for both
    // sets of limits on the side stream while the grouping runs (k_ranges_segment), and picked afterwards.
    CapChoice caps;
    caps.entries_plain = pk.stage_masks ? geo.cap_entries_masks : geo.cap_entries_plain;
    caps.loci_plain = pk.stage_masks ? geo.cap_loci_masks : geo.cap_loci_plain;
    caps.entries_counts = geo.cap_entries_counts;
    caps.loci_counts = geo.cap_loci_counts;
    caps.count_limit = kCountTileLimit;
    caps.allow_counts = (allow_count_tile && !pk.stage_masks) ? 1u : 0u;
    // Segment length of the count tile's ranges. A segment is cut greedily -- full ranges, then a remainder --, and
    // every range costs a workgroup the same set-up and the same item slots whatever it holds: a segment that
    // needs 1.3 ranges (C5: 8190 loci x 1.3 entries per cell block and locus against 8192 staged entries) becomes
    // one full range and one at 29 %, 49 ranges where 34 would do. Where a full-length segment overflows the
    // entry limit, shorter segments that hold ONE range each at 93 % (the margin covers the densest block's
    // fluctuation; a segment that overflows all the same is cut in two) are taken if they give fewer ranges.
    if (caps.allow_counts && n_kept && L) {
        const double density = (double)n_kept / ((double)num_cells / B) / (double)L;  // entries per (full cell block, locus)
        const double per_segment = density * caps.loci_counts / caps.entries_counts;  // ranges a full-length segment needs
        if (per_segment > 1.0) {
            const double greedy = std::ceil((double)L / caps.loci_counts) * std::ceil(per_segment);
            const uint32_t one = std::max(64u, (uint32_t)(0.93 * caps.entries_counts / density));
            if (one < caps.loci_counts && std::ceil((double)L / one) * 1.05 < greedy) caps.loci_counts = one;
        }
    }
    const uint32_t lo_loci = caps.allow_counts ? std::min(caps.loci_plain, caps.loci_counts) : caps.loci_plain;
    const uint32_t hi_loci = caps.allow_counts ? std::max(caps.loci_plain, caps.loci_counts) : caps.loci_plain;
    const uint32_t n_seg = (L + lo_loci - 1) / lo_loci;
    const size_t variant_stride = (size_t)n_seg * hi_loci + n_seg + 2;
    HIP_OK(S[SEGS].ensure(2 * variant_stride * 4));
    uint32_t *seg_count = S[SEGS].as<uint32_t>();  // (not over entry -> locus: the binning passes read it)
    uint32_t *seg_ends = seg_count + n_seg;
    auto cut_ranges_on_side = [&]() -> std::string {  // call when blk_off is complete on `stream`
        HIP_OK(hipEventRecord(pk.ev_offsets, stream));
        HIP_OK(hipStreamWaitEvent(pk.side, pk.ev_offsets, 0));
        hipLaunchKernelGGL(k_ranges_segment, dim3(n_seg, caps.allow_counts ? 2u : 1u), dim3(TPB), 0, pk.side, blk_off, nb,
                           L, caps, seg_ends, seg_count, variant_stride);
        HIP_OK(hipEventRecord(pk.ev_join, pk.side));
        return std::string();
    };
    if (force_radix) HIP_OK(hipMemsetAsync(blk_cnt, 0, (n_off + 1) * 4, stream));
    HIP_OK(hipMemsetAsync(per_cell_sq, 0, ((size_t)nb * B + 1) * 8, stream));
    // per kept entry what k_records needs of its read: appearance rank (the duplicate-rule flags in WORK_A
    // are dead) and base | multi flag (the grouped ids in VAL_A are dead)
    uint32_t *krank = work_a;
    uint8_t *kflags = S[VAL_A].as<uint8_t>();
    uint32_t TL = 0, locus_grid = 0;
    size_t lds = 0;
    if (!force_radix) bin_tiles(nb, &TL, &locus_grid, &lds);
    const bool hist_first = split_singles && !force_radix;  // the S entries are counted before the M entries join
    if (hist_first) {
        // (the group -> cell map rides in LDS if it fits beside the tile: 2 bytes per group)
        static const bool g2p_allowed = [] { const char *e = std::getenv("SECEDO_PACK_G2P_LDS"); return !(e && std::atoi(e) == 0); }();
        const size_t g2p_bytes = ((size_t)in.n_groups * 2 + 15) / 16 * 16;
        const bool g2p_lds = g2p_allowed && num_cells <= 0xFFFFu && lds + g2p_bytes <= 49152;
        hipLaunchKernelGGL(k_bin_hist, dim3(g2p_lds ? std::min<uint32_t>(locus_grid, 1024) : locus_grid), dim3(TPB),
                           lds + (g2p_lds ? g2p_bytes : 0), stream, raw, nb, TL, entry_kc, m_idx, num_cells, B_log2, blk_cnt,
                           sc, g2p_lds ? 1u : 0u);
    }
    if (n_m)
        hipLaunchKernelGGL(k_keys2, dim3(blocks_for(n_m)), dim3(TPB), 0, stream, sub, sval, incl, read_locus, run_rank,
                           read_off, num_cells, B, lbits, force_radix ? key2_a : nullptr, force_radix ? val2_a : nullptr,
                           t_read, krank, kflags, entry_kc, split_singles ? m_entry : nullptr,
                           hist_first ? blk_cnt : nullptr, L, sc);
    trace.mark("k_keys2 launched");
    const uint32_t slice_grid = std::min<uint32_t>(2048, (n_kept + 4095) / 4096);
    if (force_radix) {
        if (n_kept) {
            cub_cap = S[CUB].bytes;
            HIP_OK(hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_cap, key2_a, key2_b, val2_a, val2_b, (int)n_kept,
                                                       0, (int)(kCibBits + lbits) + bits_for(nb - 1), stream));
            hipLaunchKernelGGL(k_group_counts, dim3(slice_grid), dim3(TPB), 0, stream, key2_b, n_kept, B, L, lbits,
                               blk_cnt, per_cell_sq);
        }
        cub_cap = S[CUB].bytes;
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_cap, blk_cnt, blk_off, (int)n_off, stream));
        const std::string err = cut_ranges_on_side();
        if (!err.empty()) return err;
    } else {
        if (!hist_first)
            hipLaunchKernelGGL(k_bin_hist, dim3(locus_grid), dim3(TPB), lds, stream, raw, nb, TL, entry_kc, nullptr, num_cells,
                               B_log2, blk_cnt, sc, 0u);
        trace.mark("k_bin_hist launched");
        // Where most kept entries belong to multi-entry reads (clustered loci) the M entries' records are the longest
        // kernel of the packing (C3 clustered: 1.5 ms) and nothing but k_keys2 and the flush chain stands before
        // them: they start on the side stream as soon as k_keys2 is through -- beside the histogram, the offset scan
        // AND the placing pass, not behind the range cutting (round 4: 0.5 ms of a 7.6 ms packing) --, and the short
        // range cutting stays on the main stream in front of the placing pass. On sparse loci (few M entries) the
        // order of round 3: range cutting, then the records, both on the side stream.
        const bool records_first = kept_m && (uint64_t)kept_m * 4u > (uint64_t)n_kept;
        if (records_first) {
            HIP_OK(hipEventRecord(pk.ev_offsets, stream));  // k_keys2 is through
            HIP_OK(hipStreamWaitEvent(pk.side, pk.ev_offsets, 0));
            hipLaunchKernelGGL(k_m_records, dim3(blocks_for(kept_m)), dim3(TPB), 0, pk.side, raw, kept_m, t_read, read_off,
                               read_locus, read_base, krank, kflags, rbeg, flushed, m_rec, sc);
            HIP_OK(hipEventRecord(pk.ev_join, pk.side));
        }
        cub_cap = S[CUB].bytes;
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_cap, blk_cnt, blk_off, (int)n_off, stream));
        trace.mark("offset scan launched");
        if (records_first) {
            hipLaunchKernelGGL(k_ranges_segment, dim3(n_seg, caps.allow_counts ? 2u : 1u), dim3(TPB), 0, stream, blk_off, nb,
                               L, caps, seg_ends, seg_count, variant_stride);
        } else {
            const std::string err = cut_ranges_on_side();
            if (!err.empty()) return err;
            if (kept_m) {  // behind the flush chain and the range cutting, beside the placing pass
                hipLaunchKernelGGL(k_m_records, dim3(blocks_for(kept_m)), dim3(TPB), 0, pk.side, raw, kept_m, t_read,
                                   read_off, read_locus, read_base, krank, kflags, rbeg, flushed, m_rec, sc);
                HIP_OK(hipEventRecord(pk.ev_join, pk.side));
            }
        }
        if (n_kept) {
            unsigned long long *grouped = key_b;  // the sorted entry keys are dead after k_dup_rule
            HIP_OK(hipStreamWaitEvent(stream, pk.ev_flush, 0));  // the S entries' tail flags
            // (a bitmap over the cells per wave beside the cursors, if 48 KiB of LDS hold both)
            uint32_t bm_words = (nb * B + 31) / 32;
            if (lds + (size_t)(TPB / 64) * bm_words * 4 > 49152) bm_words = 0;
            hipLaunchKernelGGL(k_bin_place, dim3(locus_grid), dim3(TPB), lds + (size_t)(TPB / 64) * bm_words * 4, stream,
                               raw, nb, TL, entry_kc, blk_off, B_log2, bm_words, split_singles ? tail_begin : nullptr,
                               grouped, dupflag);
        }
    }
    // the locus ranges were cut for both sets of limits on the side stream (after the flush chain): pick
    HIP_OK(hipStreamWaitEvent(stream, pk.ev_join, 0));
    side_join.joined = true;
    if (force_radix) {
        hipLaunchKernelGGL(k_ranges_compact, dim3(1), dim3(TPB), 0, stream, seg_ends, seg_count, variant_stride, n_seg,
                           caps, pk.range_off.as<uint32_t>(), per_cell_sq, n_kept ? nb * B : 0u, -1, sc);
        if (n_kept) {
            hipLaunchKernelGGL(k_locus_info, dim3(blocks_for(L)), dim3(TPB), 0, stream, raw, pk.range_off.as<uint32_t>(),
                               sc, nullptr, locus_chr, locus_rel);
            hipLaunchKernelGGL(k_records, dim3(blocks_for(n_kept)), dim3(TPB), 0, stream, raw, key2_b, val2_b, n_kept,
                               t_read, read_off, read_locus, read_base, krank, kflags, rbeg, flushed, locus_chr,
                               locus_rel, B, lbits, pk.entry.as<uint4>(), pk.entry32.as<uint32_t>(),
                               pk.stage_masks ? pk.mask32.as<uint32_t>() : nullptr, pk.entry_read.as<uint32_t>(), sc);
        }
    } else {
        // Counting path: the entries stay in k_bin_place's order and one pass, a thread per (block, locus)
        // group, writes the records and sums the per-cell squares. The locus ranges (their range-relative
        // locus is part of a record) are cut for the count tile if it is allowed at all; the pair bound comes
        // out of the same pass, and if it forbids the count tile the ranges are cut again and the records
        // patched (deep pileups only).
        const int assumed = caps.allow_counts ? 1 : 0;
        hipLaunchKernelGGL(k_ranges_compact, dim3(1), dim3(TPB), 0, stream, seg_ends, seg_count, variant_stride, n_seg,
                           caps, pk.range_off.as<uint32_t>(), per_cell_sq, 0u, assumed, sc);
        if (n_kept) {
            hipLaunchKernelGGL(k_locus_info, dim3(blocks_for(L)), dim3(TPB), 0, stream, raw, pk.range_off.as<uint32_t>(),
                               sc, dupflag, locus_chr, locus_rel);
            const RecordTables tables{m_rec, locus_rel, pk.entry.as<uint4>(), pk.entry32.as<uint32_t>(),
                                      pk.stage_masks ? pk.mask32.as<uint32_t>() : nullptr,
                                      pk.entry_read.as<uint32_t>()};
            const unsigned long long *grouped = key_b;
            hipLaunchKernelGGL(k_entry_records, dim3(std::min<uint32_t>(1u << 16, (n_kept + 4095) / 4096)), dim3(TPB_REC),
                               0, stream, grouped, blk_off, n_kept, nb, L, B, lbits, tables, per_cell_sq, sc);
        }
        hipLaunchKernelGGL(k_pair_bound, dim3(1), dim3(TPB), 0, stream, per_cell_sq, n_kept ? nb * B : 0u, caps,
                           (uint32_t)assumed, sc);
    }
    // read-back 3: errors of the group mapping, pair bound (-> tile variant), number of ranges
    trace.mark("records launched");
    HIP_OK(read_scalars(pk, stream, sc, nullptr, &hsc, nullptr));
    trace.mark("read-back 3 arrived");
    if (!force_radix && hsc.caps_wrong && !hsc.regroup && hsc.error == 0) {
        hipLaunchKernelGGL(k_ranges_compact, dim3(1), dim3(TPB), 0, stream, seg_ends, seg_count, variant_stride, n_seg,
                           caps, pk.range_off.as<uint32_t>(), per_cell_sq, 0u, 0, sc);
        if (n_kept) {
            hipLaunchKernelGGL(k_locus_info, dim3(blocks_for(L)), dim3(TPB), 0, stream, raw, pk.range_off.as<uint32_t>(),
                               sc, nullptr, locus_chr, locus_rel);
            hipLaunchKernelGGL(k_fix_locus_rel, dim3(blocks_for(n_off)), dim3(TPB), 0, stream, blk_off, nb, L, locus_rel,
                               pk.entry32.as<uint32_t>());
        }
        HIP_OK(read_scalars(pk, stream, sc, nullptr, &hsc, nullptr));  // the number of ranges
    }
    if (hsc.error == 1) return "group id outside group_id_to_pos";
    if (hsc.error == 2) return "group_id_to_pos maps outside the matrix";
    if (hsc.regroup) {
        *retry = kRetryRadix;
        return std::string();
    }
    pk.pair_bound = hsc.pair_bound;
    pk.cell_sq = per_cell_sq;
    pk.cell_sq_n = n_kept ? nb * B : 0u;
    pk.cell_sq_host.clear();
    pk.count_tile = caps.allow_counts && pk.pair_bound < kCountTileLimit;
    pk.cap_entries = pk.count_tile ? caps.entries_counts : caps.entries_plain;
    pk.cap_loci = pk.count_tile ? caps.loci_counts : caps.loci_plain;
    pk.num_ranges = hsc.num_ranges;
    pk.n_wide = hsc.n_wide;
    HIP_OK(hipGetLastError());
    return std::string();
}

}  // namespace

std::string pack_pileup_device(const DeviceFlatPileup &in, uint32_t num_cells, uint32_t mfl,
                               uint32_t num_threads, uint32_t block_cells,
                               StageGeometry (*geometry)(uint32_t), bool allow_count_tile,
                               hipStream_t stream, DevicePacked *out, bool *need_host) {
    *need_host = false;
    if ((in.id_base16 != nullptr) == (in.id_base32 != nullptr))
        return "exactly one of id_base16 / id_base32 must be given";
    if (num_threads == 0) return "num_threads must be positive";
    if (num_cells == 0 || num_cells > 65535) return "num_cells must be in [1, 65535]";
    if (block_cells != 0 && block_cells != 64 && block_cells != 128) return "block_cells must be 0, 64 or 128";
    const uint64_t E64 = in.n_entries;
    const uint32_t L = in.n_loci;
    const size_t n_off_max = (size_t)((num_cells + 63) / 64) * ((size_t)L + 1);
    if (E64 >= (1ull << 31) || L == 0 || L >= (1u << 30) || E64 == 0 || (uint64_t)4 * num_threads > 0xFFFFFFFFull
        || n_off_max >= (1ull << 31)) {
        *need_host = true;  // sizes this path does not cover (incl. the empty pileup)
        return std::string();
    }
    bool force_radix = false;
    if (const char *env = std::getenv("SECEDO_PACK_GROUPING")) force_radix = std::string(env) == "radix";
    // an attempt can ask to be repeated: with the radix sorts (a group too long for the counting scheme),
    // or as it was but without assuming the previous call's sizes
    int retry = kNoRetry;
    bool no_assumptions = false;
    std::string err;
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (poison_level() >= 2) {  // debugging aid: nothing may depend on what the buffers held
            if (hipStreamSynchronize(stream) != hipSuccess || (out->side && hipStreamSynchronize(out->side) != hipSuccess))
                return "poison: synchronise failed";
            DeviceArena *outputs[] = {&out->blk_off, &out->entry32, &out->mask32, &out->entry,
                                      &out->entry_read, &out->range_off, &out->read_off, &out->read_locus, &out->read_base};
            for (DeviceArena *a : outputs)
                if (a->p && hipMemset(a->p, 0xA5, a->bytes) != hipSuccess) return "poison: memset failed";
            for (DeviceArena &a : out->scratch)
                if (a.p && hipMemset(a.p, 0xA5, a.bytes) != hipSuccess) return "poison: memset failed";
        }
        err = pack_attempt(in, num_cells, mfl, num_threads, block_cells, geometry, allow_count_tile, force_radix,
                           no_assumptions, stream, out, need_host, &retry);
        if (!err.empty() || retry == kNoRetry) break;
        if (retry == kRetryRadix) force_radix = true;
        if (retry == kRetrySameScheme) no_assumptions = true;
    }
    if (err.empty() && retry != kNoRetry) *need_host = true;  // cannot happen: three attempts cover both requests
    return err;
}

}  // namespace secedo
