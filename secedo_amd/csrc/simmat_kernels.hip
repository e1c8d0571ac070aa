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
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace secedo {

uint32_t counts_split(uint32_t n_tiles);

namespace {

constexpr uint32_t META_PREV_OVF = 1u << 19;
constexpr uint32_t META_NEXT_OVF = 1u << 20;
constexpr uint32_t C_CELL = 127u;
constexpr uint32_t C_BASE_SHIFT = 7;
constexpr uint32_t C_TAIL = 1u << 9;
constexpr uint32_t C_MULTI = 1u << 10;
constexpr uint32_t C_WIDE = 1u << 11;
constexpr int LUT_DIM = 129;  // llr_table.hpp: kLlrTableDim
constexpr int SLUT_DIM = 10;  // LDS copy of the table for the 8-locus windows: x_s + x_d <= 9
constexpr long long NO_PAIR = (long long)0x8000000000000000ull;  // "another locus owns this pair"

__device__ __forceinline__ double log_add(double a, double b) {
    const double hi = fmax(a, b), lo = fmin(a, b);
    return hi + log1p(exp(lo - hi));
}

#ifdef SECEDO_STAMPS
// diagnostic build only: cycle stamps around the phases of a wave batch (MI355X guide, "In-kernel
// stamps"); sums leave through counters[2..], which nothing else reads
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long stamp_real() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define STAMP(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = stamp(); __builtin_amdgcn_sched_barrier(0)
#else
#define STAMP(var)
#endif

// wave64 inclusive prefix sum in registers (DPP row shifts + row broadcasts; no LDS traffic)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2, 3
    return v;
}

// D(x_s, x_d) outside the table (a read pair that shares more than 128 loci): noted for the host, which evaluates
// it as the reference does (its wrapped integer sums, llr_table.hpp: reference_llr_any) and adds it afterwards --
// the caller adds 0 here; without a list (SECEDO_LLR_EXACT) the closed form in log space. cell_a / cell_b: the
// matrix rows of the two reads.
constexpr uint32_t REF_TABLE_MAX = 128;  // llr_table.hpp: kLlrRefMax
__device__ __noinline__ long long llr_fixed_device(const SlowPathArgs *sp, uint32_t xs, uint32_t xd, uint32_t cell_a,
                                                   uint32_t cell_b) {
    if (sp->beyond_list) {
        const uint32_t k = atomicAdd(sp->beyond_count, 1u);
        if (k < sp->beyond_cap) sp->beyond_list[k] = make_uint4(cell_a, cell_b, xs, xd);
        return 0ll;
    }
    const LlrModelDev m = sp->model;
    const int scale_log2 = sp->scale_log2;
    const double s = xs, d = xd;
    const double diff = log_add(s * m.ln_u1 + d * m.ln_v1, s * m.ln_u2 + d * m.ln_v2);
    const double same = log_add(s * m.ln_w1 + d * m.ln_z1, s * m.ln_w2 + d * m.ln_z2);
    return llrint(ldexp(diff - same, scale_log2));
}

// Slow path (a 16-locus window overflowed on the same side for both reads): merge-walk the two
// reads' kept entries (reference: similarity_matrix.cpp:223-229). Returns the fixed-point D if
// `locus` is the first locus the two reads share (only then does this incidence own the pair),
// else NO_PAIR. (Values are returned, not written through pointers: an address-taken local in
// the caller would live in scratch memory.)
__device__ __noinline__ long long slow_pair(const SlowPathArgs *sp, uint32_t r1, uint32_t r2,
                                            uint32_t locus, uint32_t cell_a, uint32_t cell_b) {
    const SlowPathArgs a = *sp;
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
    if (first != locus) return NO_PAIR;
    return xs + xd <= REF_TABLE_MAX ? a.lut[xs * LUT_DIM + xd] : llr_fixed_device(sp, xs, xd, cell_a, cell_b);
}

// One (read pair, shared locus) incidence from the full 16-byte entries (pack_host.hpp: Entry);
// g1/g2 are their global entry indices. The fixed-point log-likelihood ratio if this incidence
// owns its read pair (no earlier locus shared), else NO_PAIR.
__device__ __noinline__ long long pair_value_full(const SlowPathArgs *sp, uint32_t g1, uint32_t g2) {
    const uint4 *entry = sp->entry;
    const uint4 A1 = entry[g1], A2 = entry[g2];
    if (A1.y & A2.y & 0xFFFFu) return NO_PAIR;
    const bool same = (((A1.x ^ A2.x) >> 16) & 3u) == 0u;
    if ((A1.x & A2.x & (META_PREV_OVF | META_NEXT_OVF)) == 0u) {
        const uint32_t shared = (A1.y & A2.y) >> 16;
        const uint32_t x = A1.z ^ A2.z;
        const uint32_t diff = ((x & 0xFFFFu) | (x >> 16)) & shared;
        const uint32_t nd = __popc(diff);
        const uint32_t xd = nd + (same ? 0u : 1u);
        const uint32_t xs = __popc(shared) - nd + (same ? 1u : 0u);
        return sp->lut[xs * LUT_DIM + xd];
    }
    const uint32_t *entry_read = sp->entry_read;
    return slow_pair(sp, entry_read[g1], entry_read[g2], A1.w, A1.x & 0xFFFFu, A2.x & 0xFFFFu);
}

// Capacity of the per-wave list of deferred joint pairs (see accumulate_tiles): 192 where the LDS
// budget allows, else 0 (the batch is rescanned for its joint pairs instead).
template <int B, int THREADS, int CAPJ, int CAPL, int HCAP, bool MASKS, bool COUNTS>
constexpr int joint_list_cap() {
    if (MASKS) return 0;
    constexpr size_t fixed = (size_t)B * B * (COUNTS ? 4 : 8) + (size_t)CAPJ * 2 + ((size_t)CAPL + 2) * 2 + 16
            + (size_t)(THREADS / 64) * ((size_t)HCAP + 64 * 8);
    return fixed + (size_t)(THREADS / 64) * 192 * 10 <= 160 * 1024 ? 192 : 0;
}

// MASKS:  the 8-locus window masks are staged too, so joint (x_s, x_d) terms of multi-locus read
//         pairs are evaluated from LDS (clustered loci); otherwise such pairs go through the full
//         entries in HBM (they are rare when loci are sparse).
// COUNTS: the tile holds two 16-bit pair counters per cell pair (matching | mismatching single-
//         locus pairs, one ds_add_u32 each) instead of an int64 sum; the workgroup converts them
//         to fixed point when it flushes (exact integer arithmetic, so the result is bit-identical
//         to the int64 tile). Needs < 65536 pairs per cell pair, which the host guarantees from
//         the pileup's pair bound; joint terms bypass the tile (global atomics).
// HCAP:   size of the per-wave strip in which a batch's pairs are flattened over the lanes (below);
//         deeper batches are flattened HCAP pairs at a time.
//
// Work distribution. A workgroup walks its locus ranges; per range the column side (block J) is
// staged in LDS. Each wave then pulls batches of 64 consecutive row-side entries (block I). A lane's
// entry pairs with the c = j1 - j0 column entries of its locus, and c varies from lane to lane, so
// the batch's pairs are FLATTENED: a wave prefix sum over c gives every entry a slice [P, P + c) of
// the batch's T pairs, the slice is filled with the owning lane's number in a per-wave LDS strip,
// and pair p is then handled by lane p % 64 -- every lane busy until the batch's last 64 pairs.
template <int B, int THREADS, int CAPJ, int CAPL, int HCAP, bool MASKS, bool COUNTS>
__global__ __launch_bounds__(THREADS) void accumulate_tiles(const AccumulateArgs a) {
    static_assert(!(MASKS && COUNTS), "the count tile is for the sparse-loci variant");
    constexpr size_t TILE_BYTES = (size_t)B * B * (COUNTS ? 4 : 8);
    constexpr int WAVES = THREADS / 64;
    constexpr int JPT = (CAPJ + THREADS - 1) / THREADS;      // staged column entries per thread
    constexpr int OPT = (CAPL + 1 + THREADS - 1) / THREADS;  // staged offsets per thread
    constexpr int MCAP = joint_list_cap<B, THREADS, CAPJ, CAPL, HCAP, MASKS, COUNTS>();
    constexpr size_t WAVE_BYTES = (size_t)HCAP + 64 * 8 + (MASKS ? 64 * 4 : 0) + (size_t)MCAP * 10;

    // LDS: [ tile | sJ CAPJ u16 | sOff CAPL+2 u16 | sJm CAPJ u32, sLut (MASKS) | s_next | per wave:
    //        owner HCAP u8, wrec 64 x {entry, j0 - P}, wm 64 u32 (MASKS),
    //        mlist MCAP x {g1, g2}, mcell MCAP u16 (sparse-loci variants: deferred joint pairs) ]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long *tile64 = reinterpret_cast<unsigned long long *>(lds_raw);
    uint32_t *tile32 = reinterpret_cast<uint32_t *>(lds_raw);
    uint16_t *sJ = reinterpret_cast<uint16_t *>(lds_raw + TILE_BYTES);
    uint16_t *sOff = sJ + CAPJ;
    uint32_t *sJm = reinterpret_cast<uint32_t *>(sOff + CAPL + 2);
    long long *sLut = reinterpret_cast<long long *>(sJm + (MASKS ? CAPJ : 0));
    uint32_t *s_next = reinterpret_cast<uint32_t *>(sLut + (MASKS ? SLUT_DIM * SLUT_DIM : 0));
    unsigned char *wave_base = reinterpret_cast<unsigned char *>(s_next + 4) + (threadIdx.x >> 6) * WAVE_BYTES;
    unsigned char *owner = wave_base;
    uint2 *wrec = reinterpret_cast<uint2 *>(wave_base + HCAP);
    uint32_t *wm = reinterpret_cast<uint32_t *>(wave_base + HCAP + 64 * 8);
    uint2 *mlist = reinterpret_cast<uint2 *>(wave_base + HCAP + 64 * 8);  // MASKS has no list
    uint16_t *mcell = reinterpret_cast<uint16_t *>(wave_base + HCAP + 64 * 8 + (size_t)MCAP * 8);

    // workgroup -> (tile, chunk of the tile's locus ranges); diagonal tiles hold half the pairs and
    // get half the chunks, so that all workgroups of a launch carry about the same work
    const uint32_t t_local = a.wg_tile[blockIdx.x];
    const uint32_t t = a.tile_ids ? a.tile_ids[t_local] : a.tile_begin + t_local;
    const uint32_t chunk = blockIdx.x - a.tile_wg_begin[t_local];
    const uint32_t n_chunks_t = a.tile_wg_begin[t_local + 1] - a.tile_wg_begin[t_local];
    const uint32_t I = a.tile_row[t], J = a.tile_col[t];
    const bool diag = (I == J);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t *offI = a.blk_off + (size_t)I * a.stride;
    const uint32_t *offJ = a.blk_off + (size_t)J * a.stride;
    // A tile's chunks are equal shares of its row-side entries (the pair work per row entry is even
    // along the genome); a chunk walks the locus ranges its share touches and, in the first and the
    // last of them, only its own row entries -- chunk boundaries need not be range boundaries.
    uint32_t r_begin = 0, r_end = a.num_ranges;
    uint32_t row_begin = 0u, row_end = 0xFFFFFFFFu;
    if (n_chunks_t > 1u) {
        const uint32_t L = a.stride - 1u;
        const uint32_t e0 = offI[0];
        const unsigned long long n_row = offI[L] - e0;
        row_begin = e0 + (uint32_t)(n_row * chunk / n_chunks_t);
        row_end = e0 + (uint32_t)(n_row * (chunk + 1u) / n_chunks_t);
        uint32_t before = 0, upto = 0;
        for (uint32_t base = 0; base < a.num_ranges; base += THREADS) {
            const uint32_t k = base + tid;
            bool ends_before = false, begins_inside = false;
            if (k < a.num_ranges) {
                ends_before = offI[a.range_off[k + 1u]] <= row_begin;
                begins_inside = offI[a.range_off[k]] < row_end;
            }
            before += (uint32_t)__syncthreads_count(ends_before);
            upto += (uint32_t)__syncthreads_count(begins_inside);
        }
        r_begin = __builtin_amdgcn_readfirstlane(before);
        r_end = __builtin_amdgcn_readfirstlane(upto);
        row_begin = __builtin_amdgcn_readfirstlane(row_begin);
        row_end = __builtin_amdgcn_readfirstlane(row_end);
    }

    if (COUNTS) {
        for (uint32_t i = tid; i < B * B; i += THREADS) tile32[i] = 0u;
    } else {
        for (uint32_t i = tid; i < B * B; i += THREADS) tile64[i] = 0ull;
    }
    if (MASKS) {
        for (uint32_t i = tid; i < SLUT_DIM * SLUT_DIM; i += THREADS)
            sLut[i] = a.lut[(i / SLUT_DIM) * LUT_DIM + (i % SLUT_DIM)];
    }

    const long long d10 = a.lut[1 * LUT_DIM + 0], d01 = a.lut[0 * LUT_DIM + 1];
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.acc) + (size_t)t * B * B;
    unsigned long long n_updates = 0, n_pairs = 0;
#ifdef SECEDO_STAMPS
    unsigned long long st_setup = 0, st_fill = 0, st_trip = 0, st_batches = 0, st_trips = 0, st_bar1 = 0, st_stage = 0, st_pref = 0, st_post = 0, st_t3 = 0, st_loop = 0;
    const unsigned long long st_begin = stamp();
    const unsigned long long st_real_begin = stamp_real();
#endif
    uint32_t upd = 0, skipped = 0;  // per range, 32-bit, per lane
    uint32_t n_list = 0;            // deferred joint pairs in this wave's list (wave-uniform)
    uint32_t skipped_list = 0;      // per lane, over the whole kernel
    // Joint pairs (both reads cover further loci) are rare when loci are sparse, and evaluating one
    // means dependent HBM reads: the hot loop only appends them to a per-wave list, and the list is
    // worked off 64 pairs at a time -- every lane busy, one memory latency for 64 pairs.
    auto flush_list = [&]() {
        if (a.debug & 16u) { n_list = 0; return; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q = lane; q < n_list; q += 64u) {
            const uint2 g = mlist[q];
            const long long v = pair_value_full(a.slow, g.x, g.y);
            if (v == NO_PAIR) ++skipped_list;
            else if (COUNTS) atomicAdd(&dst[mcell[q]], (unsigned long long)v);
            else atomicAdd(&tile64[mcell[q]], (unsigned long long)v);
        }
        __builtin_amdgcn_wave_barrier();
        n_list = 0;
    };
    uint32_t upd_w = 0;             // per range, wave-uniform (lane 0 reports it)
    // inside a diagonal tile pairs of the same cell are skipped (:215); elsewhere cells differ
    const uint32_t cell_test = diag ? C_CELL : 0u;

    // one (read pair, shared locus) incidence: row-side entry `rec` (mask m1, global index g1)
    // against column-side entry `w` (staged index jc, global index g2)
    auto pair = [&](uint32_t rec, uint32_t m1, uint32_t w, uint32_t jc, uint32_t g1, uint32_t g2) {
        const uint32_t x = rec ^ w, both = rec & w;
        // reads that were both never flushed do not pair (:407-408)
        if ((both & C_TAIL) != 0u || ((x & cell_test) == 0u && diag)) return;
        ++upd;
        const bool differ = (x & (3u << C_BASE_SHIFT)) != 0u;
        const uint32_t cell = (rec & C_CELL) * B + (w & C_CELL);
        if (__builtin_expect((both & C_MULTI) != 0u, MASKS ? 1 : 0)) {
            // both reads cover further loci: joint (x_s, x_d) term, owned by their first shared locus
            long long v = differ ? d01 : d10;
            bool owner_here = true;
            if (MASKS && ((rec | w) & C_WIDE) == 0u) {
                const uint32_t m2 = sJm[jc];
                owner_here = (m1 & m2 & 0xFFu) == 0u;
                const uint32_t shared = ((m1 & m2) >> 8) & 0xFFu;
                if (shared) {
                    const uint32_t y = m1 ^ m2;
                    const uint32_t diff = ((y >> 16) | (y >> 24)) & shared;
                    const uint32_t nd = __popc(diff);
                    const uint32_t xd = nd + (differ ? 1u : 0u);
                    const uint32_t xs = __popc(shared) - nd + (differ ? 0u : 1u);
                    v = sLut[xs * SLUT_DIM + xd];
                }
            } else {
                v = pair_value_full(a.slow, g1, g2);
                owner_here = (v != NO_PAIR);
            }
            if (owner_here) {
                if (COUNTS) atomicAdd(&dst[cell], (unsigned long long)v);
                else atomicAdd(&tile64[cell], (unsigned long long)v);
            } else {
                ++skipped;
            }
        } else if (COUNTS) {
            atomicAdd(&tile32[cell], differ ? 0x10000u : 1u);
        } else {
            atomicAdd(&tile64[cell], (unsigned long long)(differ ? d01 : d10));
        }
    };

    // the next range's column side, in flight in registers while the current range is paired
    uint32_t pJ[JPT], pM[MASKS ? JPT : 1], pO[OPT];
    uint32_t pRec = 0, pM1 = 0;  // this wave's first row-side batch of the next range
    uint32_t n_la = 0, n_lb = 0, n_ib = 0, n_ie = 0, n_jb = 0, n_je = 0, n_dsh = 0;
    bool n_staged = false;

    // Range boundaries run ahead of the pairing so that their two dependent reads (locus span -> entry
    // offsets of the two blocks) are never waited for. One register holds them all, a value per lane:
    // lanes 0-3 the offsets of the next range (row begin, row end, column begin, column end), lanes 4-5
    // the locus span of the range after it; it is loaded while the current range is paired.
    uint32_t q_la = 0, q_lb = 0;  // locus span of the range `ahead` holds the offsets of
    uint32_t ahead = 0;
    auto fetch_ahead = [&](uint32_t r) {  // r: the range with span (q_la, q_lb)
        const uint32_t *src = lane == 0u ? offI + q_la : lane == 1u ? offI + q_lb : lane == 2u ? offJ + q_la
                            : lane == 3u ? offJ + q_lb : a.range_off + (r + lane - 3u);  // lanes 4, 5: r + 1, r + 2
        ahead = 0u;
        if (lane < 4u || (lane < 6u && r + 1u < r_end)) ahead = *src;
    };
    auto prefetch = [&](uint32_t r) {
        n_la = q_la;
        n_lb = q_lb;
        const uint32_t range_ib = __builtin_amdgcn_readlane(ahead, 0);
        n_ib = max(range_ib, row_begin);  // this chunk's part of the range's row side
        n_ie = max(n_ib, min((uint32_t)__builtin_amdgcn_readlane(ahead, 1), row_end));
        n_dsh = n_ib - range_ib;
        n_jb = __builtin_amdgcn_readlane(ahead, 2);
        n_je = __builtin_amdgcn_readlane(ahead, 3);
        q_la = __builtin_amdgcn_readlane(ahead, 4);
        q_lb = __builtin_amdgcn_readlane(ahead, 5);
        if (r + 1u < r_end) fetch_ahead(r + 1u);
        n_staged = (n_je - n_jb) <= (uint32_t)CAPJ && (n_lb - n_la) <= (uint32_t)CAPL;
        if (n_staged) {
#pragma unroll
            for (int k = 0; k < JPT; ++k) {
                const uint32_t i = tid + k * THREADS;
                if (i < n_je - n_jb) {
                    pJ[k] = a.entry32[n_jb + i];
                    if (MASKS) pM[k] = a.mask32[n_jb + i];
                }
            }
#pragma unroll
            for (int k = 0; k < OPT; ++k) {
                const uint32_t i = tid + k * THREADS;
                if (i <= n_lb - n_la) pO[k] = offJ[n_la + i];
            }
            pRec = 0;
            pM1 = 0;
            if (tid < n_ie - n_ib) {  // batch (tid >> 6), entry tid: batches 0..WAVES-1 are pre-assigned
                pRec = a.entry32[n_ib + tid];
                if (MASKS) pM1 = a.mask32[n_ib + tid];
            }
        }
    };

    if (r_begin < r_end) {
        q_la = __builtin_amdgcn_readfirstlane(a.range_off[r_begin]);
        q_lb = __builtin_amdgcn_readfirstlane(a.range_off[r_begin + 1u]);
        fetch_ahead(r_begin);
        prefetch(r_begin);
    }
    STAMP(t_pro);
    for (uint32_t r = r_begin; r < r_end; ++r) {
        const uint32_t la = n_la, lb = n_lb, ib = n_ib, ie = n_ie, jb = n_jb, je = n_je, dsh = n_dsh;
        const bool staged = n_staged;
        STAMP(tb0);
        __syncthreads();  // every wave is done with the previous range (first time: with zeroing)
        STAMP(tb1);
        if (staged) {
#pragma unroll
            for (int k = 0; k < JPT; ++k) {
                const uint32_t i = tid + k * THREADS;
                if (i < je - jb) {
                    sJ[i] = (uint16_t)pJ[k];
                    if (MASKS) sJm[i] = pM[k];
                }
            }
#pragma unroll
            for (int k = 0; k < OPT; ++k) {
                const uint32_t i = tid + k * THREADS;
                if (i <= lb - la) sOff[i] = (uint16_t)(pO[k] - jb);
            }
        }
        if (tid == 0) *s_next = WAVES;  // batches 0..WAVES-1 are pre-assigned, one per wave
        const uint32_t first_rec = pRec, first_m1 = pM1;
        __syncthreads();
        STAMP(tb2);
        if (r + 1 < r_end) prefetch(r + 1);
        STAMP(tb3);
#ifdef SECEDO_STAMPS
        st_bar1 += tb1 - tb0;
        st_stage += tb2 - tb1;
        st_pref += tb3 - tb2;
#endif

        const uint32_t nI = ie - ib;
        upd = 0;
        upd_w = 0;
        skipped = 0;
        if (staged) {
            const uint32_t n_batch = (nI + 63u) / 64u;
            uint32_t cur = tid >> 6;
            uint32_t rec = first_rec, m1 = first_m1;  // loaded while the previous range was paired
            while (cur < n_batch) {
                STAMP(t0);
                uint32_t nxt = 0;
                if (lane == 0) nxt = atomicAdd(s_next, 1u);
                nxt = __builtin_amdgcn_readfirstlane(nxt);
                uint32_t rec_n = 0, m1_n = 0;
                if (nxt < n_batch && nxt * 64u + lane < nI) {  // next batch's records in flight
                    rec_n = a.entry32[ib + nxt * 64u + lane];
                    if (MASKS) m1_n = a.mask32[ib + nxt * 64u + lane];
                }
                const uint32_t i = cur * 64u + lane;
                uint32_t j0 = 0, c = 0;
                if (i < nI) {
                    const uint32_t lrel = rec >> 16;
                    j0 = diag ? i + dsh + 1u : (uint32_t)sOff[lrel];  // diagonal: entries after this one
                    const uint32_t j1 = sOff[lrel + 1];
                    c = j1 > j0 ? j1 - j0 : 0u;
                }
                // wave inclusive prefix sum of c
                const uint32_t pin = wave_inclusive_scan(c);
                const uint32_t total = __builtin_amdgcn_readlane(pin, 63);
                const uint32_t pex = pin - c;
                if (a.debug & 8u) { cur = nxt; rec = rec_n; m1 = m1_n; continue; }
                STAMP(t1);
                if (total <= (uint32_t)HCAP) {
                    // flatten: pair p of the batch belongs to lane owner[p]
                    if (!(a.debug & 4u))
                    for (uint32_t p = pex; p < pin; ++p) owner[p] = (unsigned char)lane;
                    // what a pair needs of its row-side entry: the entry's low 16 bits, its tile row
                    // (element index) in the high 16, and j0 - P so that column index = that + p
                    wrec[lane] = make_uint2((rec & 0xFFFFu) | (((rec & C_CELL) * B) << 16), j0 - pex);
                    if (MASKS) wm[lane] = m1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    STAMP(t2);
#ifdef SECEDO_STAMPS
                    st_setup += t1 - t0;
                    st_fill += t2 - t1;
                    st_batches += 1;
                    st_trips += (total + 127u) / 128u;
#endif
                    if (!MASKS) {
                        // sparse-loci variants: single-locus pairs only in the hot loop; the (rare)
                        // pairs of two multi-locus reads are picked up by the second loop
                        // Uniform trip count, two pairs per lane and trip (independent LDS read
                        // chains), lanes past the end clamped and masked: keeps the loop free of
                        // exec-mask bookkeeping.
                        uint32_t multi_seen = 0;
                        const uint32_t last = total - 1u;
                        // One trip = PPL pairs per lane (independent LDS read chains; 4 is no faster than
                        // 2). Full trips need no bounds handling; the last trip clamps and masks, and
                        // takes one pair per lane when 64 or fewer pairs are left.
                        auto trip = [&](uint32_t base, auto ppl_c, auto tail_c) {
                            constexpr int PPL = decltype(ppl_c)::value;
                            constexpr bool TAIL = decltype(tail_c)::value;
                            if (MCAP > 0 && n_list > (uint32_t)(MCAP - 128)) flush_list();
                            uint32_t pp[PPL], oo[PPL], ww[PPL];
                            uint2 rr[PPL];
#pragma unroll
                            for (int u = 0; u < PPL; ++u) pp[u] = base + lane + 64u * u;
#pragma unroll
                            for (int u = 0; u < PPL; ++u) oo[u] = owner[TAIL ? min(pp[u], last) : pp[u]];
#pragma unroll
                            for (int u = 0; u < PPL; ++u) rr[u] = wrec[oo[u]];
#pragma unroll
                            for (int u = 0; u < PPL; ++u) ww[u] = sJ[rr[u].y + (TAIL ? min(pp[u], last) : pp[u])];
#pragma unroll
                            for (int u = 0; u < PPL; ++u) {
                                const uint32_t x = rr[u].x ^ ww[u], both = rr[u].x & ww[u];
                                // reads both never flushed do not pair (:407-408); inside a diagonal
                                // tile equal cells do not pair (:215)
                                const bool ok = (!TAIL || pp[u] < total) && (both & C_TAIL) == 0u
                                        && ((x & cell_test) != 0u || !diag);
                                upd += ok ? 1u : 0u;
                                const uint32_t cell = (rr[u].x >> 16) + (ww[u] & C_CELL);
                                if (MCAP > 0) {
                                    const bool joint = ok && (both & C_MULTI) != 0u;
                                    const unsigned long long bal = __ballot(joint);
                                    if (bal) {
                                        if (joint) {
                                            const uint32_t slot = n_list + __builtin_amdgcn_mbcnt_hi(
                                                    (uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                                            mlist[slot] = make_uint2(ib + cur * 64u + oo[u], jb + rr[u].y + pp[u]);
                                            mcell[slot] = (uint16_t)cell;
                                        }
                                        n_list += (uint32_t)__popcll(bal);
                                    }
                                } else {
                                    multi_seen |= ok ? both : 0u;
                                }
                                if ((a.debug & 1u) == 0u && ok && (both & C_MULTI) == 0u) {
                                    const bool differ = (x & (3u << C_BASE_SHIFT)) != 0u;
                                    if (COUNTS) atomicAdd(&tile32[cell], differ ? 0x10000u : 1u);
                                    else atomicAdd(&tile64[cell], (unsigned long long)(differ ? d01 : d10));
                                }
                            }
                        };
                        using two = std::integral_constant<int, 2>;
                        using one = std::integral_constant<int, 1>;
                        if (!(a.debug & 2u)) {
                            uint32_t base = 0;
                            for (; base + 128u <= total; base += 128u) trip(base, two{}, std::false_type{});
                            if (base < total) {
                                if (total - base <= 64u) trip(base, one{}, std::true_type{});
                                else trip(base, two{}, std::true_type{});
                            }
                        }
                        STAMP(t3);
#ifdef SECEDO_STAMPS
                        st_trip += t3 - t2;
                        st_t3 = t3;
#endif
                        const bool any_multi = MCAP == 0 && (multi_seen & C_MULTI) != 0u;
                        if (MCAP == 0 && __ballot(any_multi)) {
                            for (uint32_t p = lane; p < total; p += 64u) {
                                const uint32_t o = owner[p];
                                const uint2 ro = wrec[o];
                                const uint32_t jc = ro.y + p;
                                const uint32_t w = sJ[jc];
                                const uint32_t x = ro.x ^ w, both = ro.x & w;
                                const bool ok = (both & C_TAIL) == 0u && ((x & cell_test) != 0u || !diag);
                                if (ok && (both & C_MULTI) != 0u) {
                                    const long long v = pair_value_full(a.slow, ib + cur * 64u + o, jb + jc);
                                    const uint32_t cell = (ro.x >> 16) + (w & C_CELL);
                                    if (v == NO_PAIR) ++skipped;
                                    else if (COUNTS) atomicAdd(&dst[cell], (unsigned long long)v);
                                    else atomicAdd(&tile64[cell], (unsigned long long)v);
                                }
                            }
                        }
                    } else {
                        for (uint32_t p = lane; p < total; p += 64u) {
                            const uint32_t o = owner[p];
                            const uint2 ro = wrec[o];
                            const uint32_t jc = ro.y + p;
                            pair(ro.x, wm[o], sJ[jc], jc, ib + cur * 64u + o, jb + jc);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // the strip is reused by the next batch
#ifdef SECEDO_STAMPS
                    if (!MASKS) { STAMP(t4); st_post += t4 - st_t3; }
#endif
                } else {
                    // a very deep batch (more than HCAP pairs): the same flattening, HCAP pairs at a
                    // time -- pair p belongs to lane owner[p - win] -- with the general pair routine
                    wrec[lane] = make_uint2(rec & 0xFFFFu, j0 - pex);
                    if (MASKS) wm[lane] = m1;
                    for (uint32_t win = 0; win < total; win += (uint32_t)HCAP) {
                        const uint32_t wlen = min((uint32_t)HCAP, total - win);
                        const uint32_t f0 = max(pex, win), f1 = min(pin, win + (uint32_t)HCAP);
                        for (uint32_t p = f0; p < f1; ++p) owner[p - win] = (unsigned char)lane;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        for (uint32_t q = lane; q < wlen; q += 64u) {
                            const uint32_t o = owner[q];
                            const uint2 ro = wrec[o];
                            const uint32_t jc = ro.y + win + q;
                            pair(ro.x, MASKS ? wm[o] : 0u, sJ[jc], jc, ib + cur * 64u + o, jb + jc);
                        }
                        __builtin_amdgcn_wave_barrier();  // the strip is reused by the next window
                    }
                }
                cur = nxt;
                rec = rec_n;
                m1 = m1_n;
            }
        } else {
            // a locus range that does not fit the staging buffers (a single very deep locus):
            // pair it straight from HBM/L2 with the compact entries, straight into HBM; the full
            // entries are read (and exist) only for pairs of two multi-locus reads
            for (uint32_t e1 = ib + tid; e1 < ie; e1 += THREADS) {
                const uint32_t r1 = a.entry32[e1];
                const uint32_t l = la + (r1 >> 16);
                const uint32_t j0 = diag ? e1 + 1 : offJ[l];
                const uint32_t j1 = offJ[l + 1];
                const uint32_t row = (r1 & C_CELL) * B;
                for (uint32_t e2 = j0; e2 < j1; ++e2) {
                    const uint32_t r2 = a.entry32[e2];
                    const uint32_t x = r1 ^ r2, both = r1 & r2;
                    if (diag && (x & C_CELL) == 0u) continue;  // same cell (:215)
                    if (both & C_TAIL) continue;               // both never flushed (:407-408)
                    ++upd;
                    long long v = (x & (3u << C_BASE_SHIFT)) ? d01 : d10;
                    if (both & C_MULTI) {
                        v = pair_value_full(a.slow, e1, e2);
                        if (v == NO_PAIR) {
                            ++skipped;
                            continue;
                        }
                    }
                    atomicAdd(&dst[row + (r2 & C_CELL)], (unsigned long long)v);
                }
            }
        }
#ifdef SECEDO_STAMPS
        { STAMP(te); st_loop += te - tb3; }
#endif
        const uint32_t upd_all = upd + (lane == 0u ? upd_w : 0u);
        n_updates += upd_all;
        n_pairs += (unsigned long long)upd_all - skipped;
    }
    STAMP(t_le);
    if (MCAP > 0 && n_list) flush_list();
    STAMP(t_fl);
    __syncthreads();
    STAMP(t_eb);

    // flush: the workgroup's tile goes to its own slab with plain coalesced stores; reduce_slabs adds
    // the slabs of a tile into the accumulator (joint terms and deep ranges went there directly)
    if (COUNTS) {
        uint4 *out = reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(a.slab) + (size_t)blockIdx.x * B * B);
        const uint4 *src = reinterpret_cast<const uint4 *>(tile32);
        for (uint32_t i = tid; i < B * B / 4; i += THREADS) out[i] = src[i];
    } else {
        uint4 *out = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned long long *>(a.slab) + (size_t)blockIdx.x * B * B);
        const uint4 *src = reinterpret_cast<const uint4 *>(tile64);
        for (uint32_t i = tid; i < B * B / 2; i += THREADS) out[i] = src[i];
    }

    STAMP(t_sl);
#ifdef SECEDO_STAMPS
    if (threadIdx.x == 0u && blockIdx.x < 2048u) {
        // per workgroup (wave 0 stands for it), read by secedo_simmat_last_counts under SECEDO_STAMPS_PRINT:
        // where it ran (HW_REG_XCC_ID | HW_REG_HW_ID), the cycles of its phases, its real-time start and
        // end (s_memrealtime, 100 MHz), and -- in counters[16 + workgroup] -- lifetime | first range | end
        unsigned long long *w = a.counters + 16 + 2048 + (size_t)blockIdx.x * 8;
        w[0] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32)
                | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        w[1] = t_le - t_pro;
        w[2] = t_fl - t_le;
        w[3] = t_eb - t_fl;
        w[4] = t_sl - t_eb;
        w[5] = st_real_begin;
        w[6] = stamp_real();
        w[7] = st_loop;
        a.counters[16 + blockIdx.x] = ((t_sl - st_begin) & 0xFFFFFFFFull) | ((unsigned long long)r_begin << 32)
                | ((unsigned long long)r_end << 44);
    }
#endif
    // work counters: wave reduction, one atomic per wave
    n_pairs -= skipped_list;
    for (int off = 32; off > 0; off >>= 1) {
        n_updates += __shfl_down(n_updates, off);
        n_pairs += __shfl_down(n_pairs, off);
    }
    // ... and one per workgroup: thousands of same-address atomics at the end of a launch queue up in
    // the memory system and delay the workgroups still running (the staging area is free by now)
    unsigned long long *red = reinterpret_cast<unsigned long long *>(sJ);
    if (lane == 0u) {
        red[(tid >> 6) * 2] = n_updates;
        red[(tid >> 6) * 2 + 1] = n_pairs;
    }
    __syncthreads();
    if (tid == 0u) {
        unsigned long long u = 0, q = 0;
        for (int w = 0; w < WAVES; ++w) {
            u += red[w * 2];
            q += red[w * 2 + 1];
        }
        if (u | q) {
            atomicAdd(&a.counters[0], u);
            atomicAdd(&a.counters[1], q);
        }
    }
#ifdef SECEDO_STAMPS
    if (tid == 0u) {  // wave 0 of every workgroup stands for its workgroup
        atomicAdd(&a.counters[2], st_setup);
        atomicAdd(&a.counters[3], st_fill);
        atomicAdd(&a.counters[4], st_trip);
        atomicAdd(&a.counters[5], st_batches);
        atomicAdd(&a.counters[6], st_trips);
        atomicAdd(&a.counters[7], stamp() - st_begin);
        atomicAdd(&a.counters[8], 1ull);
        atomicAdd(&a.counters[9], st_bar1);
        atomicAdd(&a.counters[10], st_stage);
        atomicAdd(&a.counters[11], st_pref);
        atomicAdd(&a.counters[12], st_post);
        atomicAdd(&a.counters[13], st_loop);
        atomicMax(&a.counters[14], stamp() - st_begin);
    }
#endif
}


// ------------------------------------------------------------------------------------------------
// accumulate_counts + correct_tiles: the sparse-loci path (count tile; what C2, C3 and C5 run).
//
// accumulate_counts has the decomposition of accumulate_tiles -- a workgroup owns one B x B tile for a share
// of its row-side entries and walks the locus ranges that share touches, the column side of a range staged
// in LDS, the next range's loads in flight in registers -- but it counts EVERY (row entry, column entry)
// incidence of two different cells as a single-locus pair, whatever the reads' flags:
//
//   * ITEMS. A thread keeps its JPT row-side entries of the range in registers, each packed with what its
//     pairs need: {cell | base (9 bits), first column entry j (14), column entries c (8)}.
//   * GROUPS OF FOUR. A wave takes 64 items and pairs each with its first four column entries: four LDS
//     reads in flight at once, straight-line code, no loop, no flag test, no branch. 79 % of the slots
//     hold a pair when loci are sparse (c ~ Poisson(3.8)).
//   * CONTINUATIONS. Items with more than four column entries are pushed (ballot + mbcnt) to a per-wave
//     ring in LDS with j += 4, c -= 4, and come back as wave batches of their own -- the long tail never
//     holds up the short majority.
//   * WIDE ITEMS (c >= 32, deep loci) are paired by the whole wave, 64 column entries at a time.
//
// What the flags mean is settled afterwards by correct_tiles, per tile, over the (few) flagged entries
// only: a pair of two never-flushed reads is taken out again (:407-408), and a pair of two multi-locus
// reads that share n >= 2 loci gets D(x_s, x_d) - x_s D(1,0) - x_d D(0,1) added once, at its first shared
// locus -- all in the integer fixed point of the accumulator, so the sum is bit for bit what the joint
// evaluation in the pair loop gave. The pair loop used to append such pairs to a list and work the list
// off with dependent HBM reads: in-kernel stamps showed every range barrier waiting for whichever wave
// was flushing its list (26 % of a wave's life on C3).
// ------------------------------------------------------------------------------------------------
constexpr int COUNTS_RING = 256;     // continuation items per wave
constexpr uint32_t IT_J_SHIFT = 9, IT_C_SHIFT = 24, IT_J_MASK = 0x3FFFu, IT_WIDE = 32;
constexpr uint32_t IT_REC_MASK = 0x1FFu;  // cell (7 bits) | base (2 bits)

// GROUP: column entries an item is paired with per pass. Measured on one MI355X (accumulate phase, ms):
//   C3 (3.8 entries per cell block and locus): - / 3.31 / 3.13 / 3.02 for GROUP 1 / 2 / 3 / 4
//   C5 (1.3):                                 31.4 / 30.6 / 31.6 / -
// (with seven vector instructions per slot, before col32, the empty slots of GROUP 4 cost more than the trips
// through the ring: 4.87 / 4.49 / 4.38 / 5.62 on C3.) 4 by default, 2 below 2.5 entries per block and locus.
// SLOT_ASM: the pair slot as five hand-placed instructions between one s_and_saveexec and one s_mov exec. From the
// C++ form the compiler builds, per slot, saveexec + a branch around the (out-of-line) body + s_or exec + the test
// of the wave-uniform `diag` flag with its branch: five scalar / branch instructions for three vector ones and
// the ds_add -- 0.63e9 scalar instructions per C3 launch through the ONE scalar unit of a CU, 2.3e8 branches.
template <int B, int THREADS, int CAPJ, int CAPL, int GROUP, bool SLOT_ASM>
__global__ __launch_bounds__(THREADS) void accumulate_counts(const AccumulateArgs a) {
    static_assert(CAPJ <= 16384, "14 bits of column index in an item");
    static_assert(GROUP >= 1 && GROUP <= 8, "group size");
    // Rows of the LDS tile are B + 1 words apart: the lanes of a wave that share a locus add to the SAME column
    // (their common column entry) in DIFFERENT rows, and with a row stride of B words all of them would hit one
    // bank (bank = column mod 32): 3-4 lanes deep at every locus, on top of the random collisions.
    constexpr uint32_t ROW_WORDS = B + 1;
    constexpr size_t TILE_BYTES = ((size_t)B * ROW_WORDS * 4 + 15) / 16 * 16;
    constexpr int WAVES = THREADS / 64;
    constexpr int JPT = (CAPJ + THREADS - 1) / THREADS;      // staged / held entries per thread
    constexpr int OPT = (CAPL + 1 + THREADS - 1) / THREADS;  // staged offsets per thread
    constexpr int RING = COUNTS_RING;

    // LDS: [ tile | sJ CAPJ u32 (col32_of: column byte offset | base << 16, made of entry32 when a range is staged: a
    // copy of the entries in that form cost the packing a 4-byte store per entry) | sOff CAPL+2 u16 | per wave: ring ]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *tile32 = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *sJ = reinterpret_cast<uint32_t *>(lds_raw + TILE_BYTES);
    uint16_t *sOff = reinterpret_cast<uint16_t *>(sJ + CAPJ);
    uint32_t *ring = reinterpret_cast<uint32_t *>(sOff + CAPL + 2) + (threadIdx.x >> 6) * RING;

    const uint32_t t_local = a.wg_tile[blockIdx.x];
    const uint32_t t = a.tile_ids ? a.tile_ids[t_local] : a.tile_begin + t_local;
    const uint32_t chunk = blockIdx.x - a.tile_wg_begin[t_local];
    const uint32_t n_chunks_t = a.tile_wg_begin[t_local + 1] - a.tile_wg_begin[t_local];
    const uint32_t I = a.tile_row[t], J = a.tile_col[t];
    const bool diag = (I == J);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t *offI = a.blk_off + (size_t)I * a.stride;
    const uint32_t *offJ = a.blk_off + (size_t)J * a.stride;
    // (Handing the row entries to the threads interleaved across the waves -- so that the entries of one locus,
    // which have the same number of column entries each, land in different waves and the waves' pair counts
    // even out -- was measured slower: 3.29 against 2.88 ms on C3. The barrier wait did not shrink (it is not
    // the pair counts that differ), and the lanes of a wave lose the shared LDS reads of a common locus.)
    // chunks: equal shares of the tile's row-side entries (see accumulate_tiles)
    uint32_t r_begin = 0, r_end = a.num_ranges;
    uint32_t row_begin = 0u, row_end = 0xFFFFFFFFu;
    if (n_chunks_t > 1u) {
        const uint32_t L = a.stride - 1u;
        const uint32_t e0 = offI[0];
        const unsigned long long n_row = offI[L] - e0;
        row_begin = e0 + (uint32_t)(n_row * chunk / n_chunks_t);
        row_end = e0 + (uint32_t)(n_row * (chunk + 1u) / n_chunks_t);
        uint32_t before = 0, upto = 0;
        for (uint32_t base = 0; base < a.num_ranges; base += THREADS) {
            const uint32_t k = base + tid;
            bool ends_before = false, begins_inside = false;
            if (k < a.num_ranges) {
                ends_before = offI[a.range_off[k + 1u]] <= row_begin;
                begins_inside = offI[a.range_off[k]] < row_end;
            }
            before += (uint32_t)__syncthreads_count(ends_before);
            upto += (uint32_t)__syncthreads_count(begins_inside);
        }
        r_begin = __builtin_amdgcn_readfirstlane(before);
        r_end = __builtin_amdgcn_readfirstlane(upto);
        row_begin = __builtin_amdgcn_readfirstlane(row_begin);
        row_end = __builtin_amdgcn_readfirstlane(row_end);
    }
    for (uint32_t i = tid; i < B * ROW_WORDS; i += THREADS) tile32[i] = 0u;

    unsigned long long n_updates = 0;  // lane 0 of each wave carries the wave's count
    uint32_t upd_w = 0;                // this wave's pairs in the current range (wave-uniform)
    uint32_t ring_head = 0, ring_tail = 0;  // wave-uniform, free-running

    // the next range, in flight in registers while the current one is paired
    uint32_t pJ[JPT], pI[JPT], pO[OPT];
    uint32_t n_la = 0, n_lb = 0, n_ib = 0, n_ie = 0, n_jb = 0, n_je = 0, n_dsh = 0;
    bool n_staged = false;
    uint32_t q_la = 0, q_lb = 0;  // locus span of the range `ahead` holds the offsets of
    uint32_t ahead = 0;           // lanes 0-3: offsets of the next range; lanes 4-5: span of the one after
    auto fetch_ahead = [&](uint32_t r) {
        const uint32_t *src = lane == 0u ? offI + q_la : lane == 1u ? offI + q_lb : lane == 2u ? offJ + q_la
                            : lane == 3u ? offJ + q_lb : a.range_off + (r + lane - 3u);
        ahead = 0u;
        if (lane < 4u || (lane < 6u && r + 1u < r_end)) ahead = *src;
    };
    auto prefetch = [&](uint32_t r) {
        n_la = q_la;
        n_lb = q_lb;
        const uint32_t range_ib = __builtin_amdgcn_readlane(ahead, 0);
        n_ib = max(range_ib, row_begin);  // this chunk's part of the range's row side
        n_ie = max(n_ib, min((uint32_t)__builtin_amdgcn_readlane(ahead, 1), row_end));
        n_dsh = n_ib - range_ib;
        n_jb = __builtin_amdgcn_readlane(ahead, 2);
        n_je = __builtin_amdgcn_readlane(ahead, 3);
        q_la = __builtin_amdgcn_readlane(ahead, 4);
        q_lb = __builtin_amdgcn_readlane(ahead, 5);
        if (r + 1u < r_end) fetch_ahead(r + 1u);
        n_staged = (n_je - n_jb) <= (uint32_t)CAPJ && (n_ie - n_ib) <= (uint32_t)CAPJ
                && (n_lb - n_la) <= (uint32_t)CAPL;
        // Buffer loads: the descriptor (wave-uniform: base and byte count of the slice) does the bounds check, a
        // lane past the end gets 0 -- no compare, no exec mask, no address arithmetic per load (the per-lane
        // offset tid * 4 is loop-invariant, k * THREADS * 4 goes in the scalar offset).
        if (n_staged) {
            const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint32_t *>(a.entry32 + n_jb), 0, (int)((n_je - n_jb) * 4u), 0x00020000);
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint32_t *>(offJ + n_la), 0, (int)((n_lb - n_la + 1u) * 4u), 0x00020000);
#pragma unroll
            for (int k = 0; k < JPT; ++k)
                pJ[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rj, (int)(tid * 4u), k * THREADS * 4, 0);
#pragma unroll
            for (int k = 0; k < OPT; ++k)
                pO[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ro, (int)(tid * 4u), k * THREADS * 4, 0);
        }
    };
    // the row side of the range prefetch() described; issued later, when the registers of the current
    // range's items are free (the loads still have the ring drain and two barriers to land)
    auto prefetch_rows = [&]() {
        if (n_staged) {
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint32_t *>(a.entry32 + n_ib), 0, (int)((n_ie - n_ib) * 4u), 0x00020000);
#pragma unroll
            for (int k = 0; k < JPT; ++k)
                pI[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ri, (int)(tid * 4u), k * THREADS * 4, 0);
        }
    };

    if (r_begin < r_end) {
        q_la = __builtin_amdgcn_readfirstlane(a.range_off[r_begin]);
        q_lb = __builtin_amdgcn_readfirstlane(a.range_off[r_begin + 1u]);
        fetch_ahead(r_begin);
        prefetch(r_begin);
        prefetch_rows();
    }
#ifdef SECEDO_STAMPS
    unsigned long long st_barA = 0, st_stage = 0, st_items = 0, st_prim = 0, st_drain = 0, st_nprim = 0, st_ndrain = 0;
    const unsigned long long st_begin = stamp();
#endif
    // (Compiling the range loop twice, for diagonal tiles and for the others, to drop the run-time `diag` tests
    // from the pair slots was measured slower -- 4.26 against 3.74 ms of accumulate on C3: 128 VGPRs with a
    // spill, 9 % more vector instructions -- hence one instance with a wave-uniform flag.)
    {
        const bool DIAG = diag;
        // one (row entry, column entry) incidence per lane of `in`. w = col32_of(column entry): the address is
        // row + its low half, the base test its third byte against the row entry's base -- with sub-dword
        // operand selects three vector instructions and the ds_add
        // the tile's place in LDS as ds_add takes it (the dynamic segment starts behind whatever static LDS the
        // kernel's helpers use): folded into the row offset of the hand-placed slot
        const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds_raw);
        auto pair_slot = [&](uint32_t rec9, uint32_t row_byte, uint32_t w, unsigned long long in) {
            if (DIAG) in &= __ballot((w & 0xFFFFu) != ((rec9 & C_CELL) << 2));  // same cell (:215)
            upd_w += (uint32_t)__popcll(in);
            const uint32_t addr = row_byte + (w & 0xFFFFu);
            if (__builtin_amdgcn_inverse_ballot_w64(in))
                atomicAdd(reinterpret_cast<uint32_t *>(lds_raw + addr),
                          (w >> 16) != (rec9 >> C_BASE_SHIFT) ? 0x10000u : 1u);
        };
        // SLOT_ASM, tiles off the diagonal (all but one in num_blocks): exec = in; same base? (third byte of w
        // against the row entry's base); address = row + low half of w; 1 or 0x10000; ds_add; exec back. (The
        // s_nop covers the SDWA compare's write of vcc before v_cndmask reads it, as in the compiler's own
        // sequence. lgkmcnt: the compiler does not see this ds_add; LDS returns in order, so its counted waits
        // for earlier reads can only become stricter.)
        auto pair_slot_asm = [&](uint32_t rbase, uint32_t row_addr, uint32_t w, unsigned long long in) {
            // (the updates counted elsewhere -- per lane in the group, v_min + v_add, or as the items' column entries
            // when the items are built -- instead of this scalar population count per slot: C5 7-8 % slower both times)
            upd_w += (uint32_t)__popcll(in);
            uint32_t addr, val;
            unsigned long long saved;
            asm volatile("s_and_saveexec_b64 %[saved], %[in]\n\t"
                         "v_cmp_eq_u32_sdwa vcc, %[w], %[rb] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                         "v_add_u32_sdwa %[addr], %[row], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD "
                         "src1_sel:WORD_0\n\t"
                         "s_nop 0\n\t"
                         "v_cndmask_b32_e64 %[val], %[k], 1, vcc\n\t"
                         "ds_add_u32 %[addr], %[val]\n\t"
                         "s_mov_b64 exec, %[saved]"
                         : [saved] "=&s"(saved), [addr] "=&v"(addr), [val] "=&v"(val)
                         : [in] "s"(in), [w] "v"(w), [rb] "v"(rbase), [row] "v"(row_addr), [k] "v"(0x10000u)
                         : "vcc", "memory");
        };

        // 64 items, each against its first GROUP column entries; items with more go to the ring.
        // group_load reads the column entries, group_pair does the rest: the primary batches issue the reads
        // of the next item before they pair the current one.
        auto group_load = [&](uint32_t item, uint32_t (&w)[GROUP]) {
            const uint32_t *p = sJ + ((item >> IT_J_SHIFT) & IT_J_MASK);
            // lanes with fewer than GROUP read on inside the staging area (or the offsets behind it): harmless,
            // masked by `in`
#pragma unroll
            for (int u = 0; u < GROUP; ++u) w[u] = p[u];
        };
        auto group_pair = [&](uint32_t item, const uint32_t (&w)[GROUP]) {
            const uint32_t c = item >> IT_C_SHIFT;
            const uint32_t rec9 = item & IT_REC_MASK;
            const uint32_t row_byte = (item & C_CELL) * (ROW_WORDS * 4u);
            unsigned long long in[GROUP];
#pragma unroll
            for (int u = 0; u < GROUP; ++u) in[u] = __ballot(c > (uint32_t)u);
            const unsigned long long more = __ballot(c > (uint32_t)GROUP);
            // (the wave-uniform `diag` flag is tested once per group, not once per slot)
            if (SLOT_ASM && !DIAG && (GROUP == 4 || GROUP == 2)) {
                // The slots of a group as ONE block (round 4): the compares, addresses and values of all slots under
                // the group's exec -- a lane that does not take part computes garbage nobody adds --, each compare
                // into a scalar pair of its own, and only the ds_add under the slot's lanes. Per slot that is one
                // s_mov instead of s_and_saveexec + s_nop + s_mov: the scalar unit, which a CU has once, was busy
                // 47 % of the pair kernel, and the nop sat in front of every value (the SDWA compare's result must
                // not be read by the next instruction; here four or more instructions lie between).
                const uint32_t rbase = rec9 >> C_BASE_SHIFT, row_addr = lds_base + row_byte;
#pragma unroll
                for (int u = 0; u < GROUP; ++u) upd_w += (uint32_t)__popcll(in[u]);
                uint32_t a0, a1, a2, a3, v0, v1, v2, v3;
                unsigned long long saved, q0, q1, q2, q3;
                if (GROUP == 4) {
                    asm volatile("v_cmp_eq_u32_sdwa %[q0], %[w0], %[rb] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                                 "v_cmp_eq_u32_sdwa %[q1], %[w1], %[rb] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                                 "v_cmp_eq_u32_sdwa %[q2], %[w2], %[rb] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                                 "v_cmp_eq_u32_sdwa %[q3], %[w3], %[rb] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                                 "v_add_u32_sdwa %[a0], %[row], %[w0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                                 "v_add_u32_sdwa %[a1], %[row], %[w1] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                                 "v_add_u32_sdwa %[a2], %[row], %[w2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                                 "v_add_u32_sdwa %[a3], %[row], %[w3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                                 "v_cndmask_b32_e64 %[v0], %[k], 1, %[q0]\n\t"
                                 "v_cndmask_b32_e64 %[v1], %[k], 1, %[q1]\n\t"
                                 "v_cndmask_b32_e64 %[v2], %[k], 1, %[q2]\n\t"
                                 "v_cndmask_b32_e64 %[v3], %[k], 1, %[q3]\n\t"
                                 "s_mov_b64 %[saved], exec\n\t"
                                 "s_mov_b64 exec, %[in0]\n\t"
                                 "ds_add_u32 %[a0], %[v0]\n\t"
                                 "s_mov_b64 exec, %[in1]\n\t"
                                 "ds_add_u32 %[a1], %[v1]\n\t"
                                 "s_mov_b64 exec, %[in2]\n\t"
                                 "ds_add_u32 %[a2], %[v2]\n\t"
                                 "s_mov_b64 exec, %[in3]\n\t"
                                 "ds_add_u32 %[a3], %[v3]\n\t"
                                 "s_mov_b64 exec, %[saved]"
                                 : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [v0] "=&v"(v0), [v1] "=&v"(v1),
                                   [v2] "=&v"(v2), [v3] "=&v"(v3), [saved] "=&s"(saved), [q0] "=&s"(q0), [q1] "=&s"(q1),
                                   [q2] "=&s"(q2), [q3] "=&s"(q3)
                                 : [w0] "v"(w[0]), [w1] "v"(w[1 % GROUP]), [w2] "v"(w[2 % GROUP]), [w3] "v"(w[3 % GROUP]),
                                   [rb] "v"(rbase), [row] "v"(row_addr), [k] "v"(0x10000u), [in0] "s"(in[0]),
                                   [in1] "s"(in[1 % GROUP]), [in2] "s"(in[2 % GROUP]), [in3] "s"(in[3 % GROUP])
                                 : "memory");
                } else {
                    asm volatile("v_cmp_eq_u32_sdwa %[q0], %[w0], %[rb] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                                 "v_cmp_eq_u32_sdwa %[q1], %[w1], %[rb] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                                 "v_add_u32_sdwa %[a0], %[row], %[w0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                                 "v_add_u32_sdwa %[a1], %[row], %[w1] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                                 "s_mov_b64 %[saved], exec\n\t"
                                 "v_cndmask_b32_e64 %[v0], %[k], 1, %[q0]\n\t"
                                 "v_cndmask_b32_e64 %[v1], %[k], 1, %[q1]\n\t"
                                 "s_mov_b64 exec, %[in0]\n\t"
                                 "ds_add_u32 %[a0], %[v0]\n\t"
                                 "s_mov_b64 exec, %[in1]\n\t"
                                 "ds_add_u32 %[a1], %[v1]\n\t"
                                 "s_mov_b64 exec, %[saved]"
                                 : [a0] "=&v"(a0), [a1] "=&v"(a1), [v0] "=&v"(v0), [v1] "=&v"(v1), [saved] "=&s"(saved),
                                   [q0] "=&s"(q0), [q1] "=&s"(q1)
                                 : [w0] "v"(w[0]), [w1] "v"(w[1 % GROUP]), [rb] "v"(rbase), [row] "v"(row_addr), [k] "v"(0x10000u),
                                   [in0] "s"(in[0]), [in1] "s"(in[1 % GROUP])
                                 : "memory");
                    (void)a2; (void)a3; (void)v2; (void)v3; (void)q2; (void)q3;
                }
            } else if (SLOT_ASM && !DIAG) {
                const uint32_t rbase = rec9 >> C_BASE_SHIFT, row_addr = lds_base + row_byte;
#pragma unroll
                for (int u = 0; u < GROUP; ++u) pair_slot_asm(rbase, row_addr, w[u], in[u]);
            } else {
#pragma unroll
                for (int u = 0; u < GROUP; ++u) pair_slot(rec9, row_byte, w[u], in[u]);
            }
            if (more) {
                if (__builtin_amdgcn_inverse_ballot_w64(more)) {
                    const uint32_t slot = ring_tail + __builtin_amdgcn_mbcnt_hi(
                            (uint32_t)(more >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                    ring[slot & (RING - 1)] = item + ((uint32_t)GROUP << IT_J_SHIFT) - ((uint32_t)GROUP << IT_C_SHIFT);
                }
                ring_tail += (uint32_t)__popcll(more);
            }
        };
        auto group4 = [&](uint32_t item) {
            uint32_t w[GROUP];
            group_load(item, w);
            group_pair(item, w);
        };
        // one batch from the ring (up to 64 items)
        auto drain_one = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n = min(64u, ring_tail - ring_head);
            uint32_t it = 0u;
            if (lane < n) it = ring[(ring_head + lane) & (RING - 1)];
            __builtin_amdgcn_wave_barrier();
            ring_head += n;
            group4(it);
        };

        for (uint32_t r = r_begin; r < r_end; ++r) {
            const uint32_t la = n_la, ib = n_ib, ie = n_ie, jb = n_jb, dsh = n_dsh;
            const bool staged = n_staged;
            STAMP(s0);
            __syncthreads();  // every wave is done with the previous range (first time: with zeroing)
            STAMP(s1);
            if (staged) {
                // (slots past the slice get the zeros of the buffer loads: the staging arrays hold JPT * THREADS
                // entries and OPT * THREADS offsets, so every store is inside them, and nothing reads those slots)
                static_assert(JPT * THREADS <= CAPJ && OPT * THREADS <= CAPL + 2, "unconditional staging stores");
#pragma unroll
                for (int k = 0; k < JPT; ++k)  // the column side's form of an entry: col32_of(cell in block, base)
                    sJ[tid + k * THREADS] = ((pJ[k] & C_CELL) << 2) | ((pJ[k] << (16 - C_BASE_SHIFT)) & 0x30000u);
#pragma unroll
                for (int k = 0; k < OPT; ++k) sOff[tid + k * THREADS] = (uint16_t)(pO[k] - jb);
            }
            __syncthreads();
            STAMP(s2);

            const uint32_t nI = ie - ib;
            upd_w = 0;
            if (staged) {
                // items of this thread's row entries: the column entries of the entry's locus are
                // sJ[j0 .. j0 + c); in a diagonal tile the entries after this one (each pair once)
                uint32_t item[JPT];
                uint32_t any_wide = 0;
#pragma unroll
                for (int k = 0; k < JPT; ++k) {
                    const uint32_t i = tid + k * THREADS;
                    const uint32_t rec = pI[k];
                    const uint32_t lrel = rec >> 16;
                    uint32_t j0 = sOff[lrel];
                    const uint32_t j1 = sOff[lrel + 1];
                    if (DIAG) j0 = i + dsh + 1u;
                    const uint32_t c = (i < nI && j1 > j0) ? j1 - j0 : 0u;
                    any_wide |= c;
                    // (j0 can be CAPJ when c is 0: the last entry of a full diagonal range; c >= 256 spills into
                    // nothing: the item is rebuilt below)
                    item[k] = (rec & IT_REC_MASK) | ((j0 & IT_J_MASK) << IT_J_SHIFT) | (c << IT_C_SHIFT);
                }
                // wide entries (deep loci; none when loci are sparse: one test for the thread's JPT items): the
                // whole wave pairs one row entry with 64 column entries at a time, right here
                if (__ballot(any_wide >= IT_WIDE)) {
#pragma unroll
                    for (int k = 0; k < JPT; ++k) {
                        const uint32_t i = tid + k * THREADS;
                        const uint32_t rec = pI[k];
                        const uint32_t lrel = rec >> 16;
                        uint32_t j0 = sOff[lrel];
                        const uint32_t j1 = sOff[lrel + 1];
                        if (DIAG) j0 = i + dsh + 1u;
                        const uint32_t c = (i < nI && j1 > j0) ? j1 - j0 : 0u;
                        unsigned long long todo = __ballot(c >= IT_WIDE);
                        while (todo) {
                            const int src = __builtin_ctzll(todo);
                            todo &= todo - 1ull;
                            const uint32_t recw = __builtin_amdgcn_readlane(rec, src);
                            const uint32_t cw = __builtin_amdgcn_readlane(c, src);
                            const uint32_t j0w = __builtin_amdgcn_readlane(j0, src);
                            const uint32_t row_byte = (recw & C_CELL) * (ROW_WORDS * 4u);
                            for (uint32_t base = 0; base < cw; base += 64u) {
                                const uint32_t jj = j0w + base + lane;
                                const unsigned long long in = __ballot(base + lane < cw);
                                const uint32_t w = sJ[min(jj, (uint32_t)CAPJ - 1u)];  // (col32 form)
                                pair_slot(recw & IT_REC_MASK, row_byte, w, in);
                            }
                        }
                        if (c >= IT_WIDE) item[k] = 0u;  // done
                    }
                }
                if (r + 1 < r_end) prefetch(r + 1);  // pJ / pO are free again: the next range's column side
                STAMP(s3);
                {
                    uint32_t wc[GROUP], wn[GROUP];
                    group_load(item[0], wc);
#pragma unroll
                    for (int k = 0; k < JPT; ++k) {
                        if (k + 1 < JPT) group_load(item[k + 1], wn);
                        // room for 64 more continuations (a drained batch may push up to 64 itself)
                        while (ring_tail - ring_head > (uint32_t)(RING - 64)) drain_one();
                        group_pair(item[k], wc);
#pragma unroll
                        for (int u = 0; u < GROUP; ++u) wc[u] = wn[u];
                    }
                }
                // (the row side right after the column side, its registers being free once the items are built, was
                // measured slower: C3 2.84 against 2.77 ms, C5 28.5 against 26.5)
                if (r + 1 < r_end) prefetch_rows();
                STAMP(s4);
#ifdef SECEDO_STAMPS
                st_ndrain += (ring_tail - ring_head + 63u) / 64u;
                st_nprim += JPT;
#endif
                while (ring_tail != ring_head) drain_one();
                STAMP(s5);
#ifdef SECEDO_STAMPS
                st_barA += s1 - s0;
                st_stage += s2 - s1;
                st_items += s3 - s2;
                st_prim += s4 - s3;
                st_drain += s5 - s4;
#endif
            } else {
                if (r + 1 < r_end) {
                    prefetch(r + 1);
                    prefetch_rows();
                }
                // a locus range that does not fit the staging buffers (a single very deep locus): paired
                // straight from HBM/L2, into the same count tile (the pair bound covers these pairs too; flags
                // are settled by correct_tiles here too)
                uint32_t upd = 0;
                for (uint32_t e1 = ib + tid; e1 < ie; e1 += THREADS) {
                    const uint32_t r1 = a.entry32[e1];
                    const uint32_t l = la + (r1 >> 16);
                    const uint32_t j0 = DIAG ? e1 + 1 : offJ[l];
                    const uint32_t j1 = offJ[l + 1];
                    const uint32_t row = (r1 & C_CELL) * ROW_WORDS;
                    for (uint32_t e2 = j0; e2 < j1; ++e2) {
                        const uint32_t r2 = a.entry32[e2];
                        const uint32_t x = r1 ^ r2;
                        if (DIAG && (x & C_CELL) == 0u) continue;  // same cell (:215)
                        ++upd;
                        atomicAdd(&tile32[row + (r2 & C_CELL)], (x & (3u << C_BASE_SHIFT)) ? 0x10000u : 1u);
                    }
                }
                n_updates += upd;
            }
            if (lane == 0u) n_updates += upd_w;
        }
    }
#ifdef SECEDO_STAMPS
    if (lane == 0u && (blockIdx.x & 15u) == 0u) {  // a sample of the workgroups, every wave of them
        atomicAdd(&a.counters[2], st_barA);
        atomicAdd(&a.counters[3], st_stage);
        atomicAdd(&a.counters[4], st_items);
        atomicAdd(&a.counters[5], st_prim);
        atomicAdd(&a.counters[6], st_drain);
        atomicAdd(&a.counters[7], stamp() - st_begin);
        atomicAdd(&a.counters[8], 1ull);
        atomicAdd(&a.counters[9], st_nprim);
        atomicAdd(&a.counters[10], st_ndrain);
        // per wave index: barrier-A wait, pair work (primary + drain), item phase
        atomicAdd(&a.counters[16 + (tid >> 6) * 4 + 0], st_barA);
        atomicAdd(&a.counters[16 + (tid >> 6) * 4 + 1], st_prim + st_drain);
        atomicAdd(&a.counters[16 + (tid >> 6) * 4 + 2], st_items);
        atomicAdd(&a.counters[16 + (tid >> 6) * 4 + 3], st_stage);
    }
#endif
    __syncthreads();

    // flush: the tile goes to the workgroup's own slab with plain coalesced stores (reduce_slabs adds up)
    {
        uint32_t *out = reinterpret_cast<uint32_t *>(a.slab) + (size_t)blockIdx.x * B * B;
        for (uint32_t i = tid; i < B * B; i += THREADS) out[i] = tile32[(i / B) * ROW_WORDS + (i % B)];  // dense rows
    }
    // work counter: wave reduction, then one atomic pair per workgroup (see accumulate_tiles); every
    // incidence counts as an update and as a read pair here, correct_tiles takes back what is neither
    for (int off = 32; off > 0; off >>= 1) n_updates += __shfl_down(n_updates, off);
    unsigned long long *red = reinterpret_cast<unsigned long long *>(sJ);
    if (lane == 0u) red[tid >> 6] = n_updates;
    __syncthreads();
    if (tid == 0u) {
        unsigned long long u = 0;
        for (int w = 0; w < WAVES; ++w) u += red[w];
        if (u) {
            atomicAdd(&a.counters[0], u);
            atomicAdd(&a.counters[1], u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate_masks: clustered loci (the 8-locus window masks staged), in the shape of accumulate_counts.
//
// accumulate_tiles<MASKS> flattens a batch's pairs over the lanes and then walks, per pair, a chain of
// dependent LDS reads (owner strip -> row record -> row mask -> column entry -> column mask -> table): one
// pair per lane and trip, nothing in flight beside it. Here the row side lives in registers as in
// accumulate_counts -- a thread keeps its JPT row entries of the range as items --, a wave pairs 64 items with
// GROUP column entries each, the column entry arrives in ONE 8-byte LDS read, issued for the next item before
// the current one is paired, and the value comes out of ONE unconditional table read. Longer items go through
// the wave's ring, 32 or more column entries are paired by the whole wave.
//
// Round 4: the words a pair is decided from. Every packed entry is converted once per prepare (masks_words) into
//   x = prev8 | tail << 8 | dead << 9 | base << 16 | (column: cell << 26; row: next8 << 24)
//   y = the bases of the read at the 8 next loci as four one-hot planes of 8 bits (plane b, bit d: the read
//       covers the locus d + 1 behind this one with base b; no bit where it does not cover it)
// so that the loci two reads share behind this one are popc(S1 & y2) (S1 = the row read's next8 in all four
// planes) and the ones with equal bases popc(y1 & y2): one AND and one population count each, where the two
// bit planes of mask32 took an XOR, two shifts, an OR, two ANDs and two counts. The base at THIS locus is a
// byte of its own in x (one byte compare), and x1 & x2 yields in one AND what ownership (prev8: an earlier
// shared locus owns the pair) and the exclusions (both reads never flushed, :407-408; a column entry whose
// read reaches beyond its windows: wide_pairs has its pairs, a row entry of that kind gets no item) need --
// masks of single-locus reads are empty, so no "both multi-locus" test is left. The table is indexed by
// (x_s, shared next loci). 47 -> 16 vector instructions per slot (by hand, pair_group3_asm); C3 clustered 47.6 -> 38.0 ms
// together with masks_words(), the two rounds in line and the spread of thin ring batches.
// A pair of two multi-locus reads is owned by its first shared locus (prev masks disjoint), as before.
// ------------------------------------------------------------------------------------------------
constexpr int MASKS_RING = 128;
constexpr uint32_t MASKS_ROW_Q(int B) { return (uint32_t)B + 1u; }
constexpr uint32_t MK_J_MASK = 0xFFFu, MK_ROW_SHIFT = 12, MK_ROW_MASK = 0x7Fu, MK_C_SHIFT = 24;
// x: [15:0] column: cell * 8 (the byte offset in a tile row), row: next8; [23:16] prev8; [24] tail; [26] dead;
//    [31:28] the base at this locus, one-hot
constexpr uint32_t MX_PREV = 0xFFu << 16, MX_TAIL = 1u << 24, MX_DEAD = 1u << 26, MX_BASE = 0xFu << 28;
constexpr int MLUT_STRIDE = 11;  // table row = x_s (0 .. 9), column = shared next loci (0 .. 8)
constexpr int MLUT_WORDS = 10 * MLUT_STRIDE;

// mask32 -> the four one-hot base planes of the 8 next loci: next8, b0, b1 each spread over the four bytes, then
// byte p keeps the loci whose (b0, b1) spell p
__device__ __forceinline__ uint32_t masks_planes(uint32_t m) {
    const uint32_t n4 = __builtin_amdgcn_perm(m, m, 0x01010101u);
    const uint32_t b0 = __builtin_amdgcn_perm(m, m, 0x02020202u) ^ 0x00FF00FFu;
    const uint32_t b1 = __builtin_amdgcn_perm(m, m, 0x03030303u) ^ 0x0000FFFFu;
    return n4 & b0 & b1;
}
// entry32 / mask32 -> x without its low half: prev8, tail (bit 9 -> 24), wide (bit 11 -> 26), the base one-hot
__device__ __forceinline__ uint32_t masks_flags(uint32_t e, uint32_t m) {
    const uint32_t base = (e >> C_BASE_SHIFT) & 3u;
    return ((m & 0xFFu) << 16) | ((e & (C_TAIL | C_WIDE)) << 15) | (0x10000000u << base);
}
static_assert((C_TAIL << 15) == MX_TAIL && (C_WIDE << 15) == MX_DEAD, "flag positions");

template <int B, int THREADS, int CAPJ, int CAPL, int GROUP>
__global__ __launch_bounds__(THREADS) void accumulate_masks(const AccumulateArgs a) {
    static_assert(CAPJ <= 4096, "12 bits of column index in an item");
        static_assert(GROUP >= 1 && GROUP <= 4, "group size");
    // rows of the int64 tile are B + 1 words apart: lanes that share a locus hold the same column entry and
    // different rows, and with a row stride of 512 bytes they all met in one pair of banks (52 % of the LDS
    // cycles were conflict cycles on C3 clustered, 22 % since); the flush takes the skew out again
    constexpr uint32_t ROW_Q = MASKS_ROW_Q(B);
    constexpr size_t TILE_BYTES = (size_t)B * ROW_Q * 8;
    constexpr int WAVES = THREADS / 64;
    constexpr int JPT = (CAPJ + THREADS - 1) / THREADS;      // staged / held entries per thread
    constexpr int OPT = (CAPL + 1 + THREADS - 1) / THREADS;  // staged offsets per thread
    constexpr int RING = MASKS_RING;

    // LDS: [ tile | sJ CAPJ x {x, y} | sOff CAPL+2 u16 | sLut | per wave: ring of {item, x1}, ring of y1 ]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long *tile64 = reinterpret_cast<unsigned long long *>(lds_raw);
    uint2 *sJ = reinterpret_cast<uint2 *>(lds_raw + TILE_BYTES);
    uint16_t *sOff = reinterpret_cast<uint16_t *>(sJ + CAPJ);
    long long *sLut = reinterpret_cast<long long *>(sOff + CAPL + 2);
    uint2 *ring = reinterpret_cast<uint2 *>(sLut + MLUT_WORDS) + (threadIdx.x >> 6) * RING;
    uint32_t *ring_y = reinterpret_cast<uint32_t *>(reinterpret_cast<uint2 *>(sLut + MLUT_WORDS) + WAVES * RING)
            + (threadIdx.x >> 6) * RING;

    const uint32_t t_local = a.wg_tile[blockIdx.x];
    const uint32_t t = a.tile_ids ? a.tile_ids[t_local] : a.tile_begin + t_local;
    const uint32_t chunk = blockIdx.x - a.tile_wg_begin[t_local];
    const uint32_t n_chunks_t = a.tile_wg_begin[t_local + 1] - a.tile_wg_begin[t_local];
    const uint32_t I = a.tile_row[t], J = a.tile_col[t];
    const bool DIAG = (I == J);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t *offI = a.blk_off + (size_t)I * a.stride;
    const uint32_t *offJ = a.blk_off + (size_t)J * a.stride;
    // chunks: equal shares of the tile's row-side entries (see accumulate_tiles)
    uint32_t r_begin = 0, r_end = a.num_ranges;
    uint32_t row_begin = 0u, row_end = 0xFFFFFFFFu;
    if (n_chunks_t > 1u) {
        const uint32_t L = a.stride - 1u;
        const uint32_t e0 = offI[0];
        const unsigned long long n_row = offI[L] - e0;
        row_begin = e0 + (uint32_t)(n_row * chunk / n_chunks_t);
        row_end = e0 + (uint32_t)(n_row * (chunk + 1u) / n_chunks_t);
        uint32_t before = 0, upto = 0;
        for (uint32_t base = 0; base < a.num_ranges; base += THREADS) {
            const uint32_t k = base + tid;
            bool ends_before = false, begins_inside = false;
            if (k < a.num_ranges) {
                ends_before = offI[a.range_off[k + 1u]] <= row_begin;
                begins_inside = offI[a.range_off[k]] < row_end;
            }
            before += (uint32_t)__syncthreads_count(ends_before);
            upto += (uint32_t)__syncthreads_count(begins_inside);
        }
        r_begin = __builtin_amdgcn_readfirstlane(before);
        r_end = __builtin_amdgcn_readfirstlane(upto);
        row_begin = __builtin_amdgcn_readfirstlane(row_begin);
        row_end = __builtin_amdgcn_readfirstlane(row_end);
    }
    for (uint32_t i = tid; i < B * ROW_Q; i += THREADS) tile64[i] = 0ull;
    // the table by (x_s, shared next loci n): D(x_s, n + 1 - x_s); x_s > n + 1 cannot occur
    for (uint32_t i = tid; i < (uint32_t)MLUT_WORDS; i += THREADS) {
        const uint32_t xs = i / MLUT_STRIDE, n = i % MLUT_STRIDE;
        sLut[i] = (n <= 8u && xs <= n + 1u) ? a.lut[xs * LUT_DIM + (n + 1u - xs)] : 0ll;
    }

    unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.acc) + (size_t)t * B * B;
    const bool slot_asm = a.masks_slot_asm != 0u;   // (SECEDO_MASKS_SLOT_ASM=0: the compiler's slot, for A/B runs)
    unsigned long long n_updates = 0, n_pairs = 0;  // per lane
    uint32_t upd_lane = 0, add_lane = 0;            // this lane's incidences / owned incidences in the range
    uint32_t ring_head = 0, ring_tail = 0;          // wave-uniform, free-running

    // the next range, in flight in registers while the current one is paired
    uint32_t pJ[JPT], pMj[JPT], pI[JPT], pXi[JPT], pMi[JPT], pO[OPT];
    uint32_t n_la = 0, n_lb = 0, n_ib = 0, n_ie = 0, n_jb = 0, n_je = 0, n_dsh = 0;
    bool n_staged = false;
    uint32_t q_la = 0, q_lb = 0;  // locus span of the range `ahead` holds the offsets of
    uint32_t ahead = 0;           // lanes 0-3: offsets of the next range; lanes 4-5: span of the one after
    auto fetch_ahead = [&](uint32_t r) {
        const uint32_t *src = lane == 0u ? offI + q_la : lane == 1u ? offI + q_lb : lane == 2u ? offJ + q_la
                            : lane == 3u ? offJ + q_lb : a.range_off + (r + lane - 3u);
        ahead = 0u;
        if (lane < 4u || (lane < 6u && r + 1u < r_end)) ahead = *src;
    };
    auto prefetch = [&](uint32_t r) {
        n_la = q_la;
        n_lb = q_lb;
        const uint32_t range_ib = __builtin_amdgcn_readlane(ahead, 0);
        n_ib = max(range_ib, row_begin);  // this chunk's part of the range's row side
        n_ie = max(n_ib, min((uint32_t)__builtin_amdgcn_readlane(ahead, 1), row_end));
        n_dsh = n_ib - range_ib;
        n_jb = __builtin_amdgcn_readlane(ahead, 2);
        n_je = __builtin_amdgcn_readlane(ahead, 3);
        q_la = __builtin_amdgcn_readlane(ahead, 4);
        q_lb = __builtin_amdgcn_readlane(ahead, 5);
        if (r + 1u < r_end) fetch_ahead(r + 1u);
        n_staged = (n_je - n_jb) <= (uint32_t)CAPJ && (n_ie - n_ib) <= (uint32_t)CAPJ
                && (n_lb - n_la) <= (uint32_t)CAPL;
        // buffer loads: the descriptor bounds the slice, a lane past the end gets 0 (see accumulate_counts)
        if (n_staged) {
            // (the column entries' words as masks_words() made them once per prepare: x with the cell's byte offset, y)
            const int nj = (int)((n_je - n_jb) * 4u);
            const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.mk_xcol + n_jb), 0, nj, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.mk_y + n_jb), 0, nj, 0x00020000);
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint32_t *>(offJ + n_la), 0, (int)((n_lb - n_la + 1u) * 4u), 0x00020000);
#pragma unroll
            for (int k = 0; k < JPT; ++k) {
                pJ[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rj, (int)(tid * 4u), k * THREADS * 4, 0);
                pMj[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rm, (int)(tid * 4u), k * THREADS * 4, 0);
            }
#pragma unroll
            for (int k = 0; k < OPT; ++k)
                pO[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ro, (int)(tid * 4u), k * THREADS * 4, 0);
        }
    };
    auto prefetch_rows = [&]() {
        if (n_staged) {
            // (the row entries: entry32 for the locus, the cell and the wide flag, and their two words in the row form)
            const int ni = (int)((n_ie - n_ib) * 4u);
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.entry32 + n_ib), 0, ni, 0x00020000);
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.mk_xrow + n_ib), 0, ni, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.mk_y + n_ib), 0, ni, 0x00020000);
#pragma unroll
            for (int k = 0; k < JPT; ++k) {
                pI[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ri, (int)(tid * 4u), k * THREADS * 4, 0);
                pXi[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rx, (int)(tid * 4u), k * THREADS * 4, 0);
                pMi[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rm, (int)(tid * 4u), k * THREADS * 4, 0);
            }
        }
    };
    if (r_begin < r_end) {
        q_la = __builtin_amdgcn_readfirstlane(a.range_off[r_begin]);
        q_lb = __builtin_amdgcn_readfirstlane(a.range_off[r_begin + 1u]);
        fetch_ahead(r_begin);
        prefetch(r_begin);
        prefetch_rows();
    }

    // GROUP (row entry, column entry) incidences per lane: `item` = {first column entry, row cell, count}, x1 / y1
    // the row entry's words, w2[u] = the column entry's, a lane takes part in slot u iff c > u. Every condition is
    // a compare on plain words combined without short circuits (as `a && b` the compiler had wrapped them in
    // exec-masked regions), and the counters are per lane (an add with the condition as carry). Phase 1 decides
    // who pairs, who owns, and the table index; phase 2 reads the table (all GROUP reads in flight together);
    // phase 3 adds.
    auto pair_group = [&](auto diag_tag, uint32_t item, uint32_t x1, uint32_t y1, const uint2 (&w2)[GROUP], uint32_t c) {
        constexpr bool DG = decltype(diag_tag)::value;  // a diagonal tile: equal cells do not pair (:215)
        const uint32_t rcell = (item >> MK_ROW_SHIFT) & MK_ROW_MASK;
        const uint32_t row_bytes = rcell * (ROW_Q * 8u);
        const uint32_t s1 = __builtin_amdgcn_perm(x1, x1, 0u);  // the row read's next8 in all four planes
        uint32_t lut_idx[GROUP], cell_off[GROUP];
        bool add[GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const uint32_t wx = w2[u].x, wy = w2[u].y;
            const uint32_t fz = x1 & wx;
            // reads both never flushed do not pair (:407-408); a column entry whose read reaches beyond its windows
            // is left to wide_pairs (row entries of that kind have no item)
            bool act = (c > (uint32_t)u) & ((fz & (MX_TAIL | MX_DEAD)) == 0u);
            if (DG) act &= (wx & 0xFFFFu) != rcell * 8u;
            // two multi-locus reads: the pair belongs to their first shared locus; x_s over the loci they share
            // behind this one plus this locus. (A single-locus read has empty masks: it shares this locus only.)
            const bool addu = act & ((fz & MX_PREV) == 0u);
            const uint32_t n = __popc(s1 & wy);
            const uint32_t xs = __popc(y1 & wy) + __popc(fz & MX_BASE);
            lut_idx[u] = xs * (uint32_t)MLUT_STRIDE + n;  // x_s <= n + 1 <= 9: inside the table
            cell_off[u] = row_bytes + (wx & 0xFFFFu);
            add[u] = addu;
            upd_lane += act ? 1u : 0u;
            add_lane += addu ? 1u : 0u;
        }
        long long v[GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; ++u) v[u] = sLut[lut_idx[u]];
#pragma unroll
        for (int u = 0; u < GROUP; ++u)
            if (add[u])
                atomicAdd(reinterpret_cast<unsigned long long *>(lds_raw + cell_off[u]), (unsigned long long)v[u]);
    };
    // The same three slots of an off-diagonal tile (all but one in num_blocks) by hand: 15 vector instructions a
    // slot where the compiler's code above has about 30 (it rebuilds lane masks from predicates for the per-lane
    // counters and spends three instructions on each address). Phase 1, per slot: the three ANDs, the counts,
    // x_s * 11 + n, the table read -- all three reads in flight; phase 2, per slot: who pairs (no tail / dead bit:
    // the masked word below 1 << 24) and who owns (the masked word zero) as two compares of ONE masked word,
    // the per-lane counters as adds with the condition as carry, the tile address as row + low half of the
    // column word (SDWA), the add under exec = owners. lgkmcnt: LDS returns in order, so after the three reads
    // "at most two outstanding" means the oldest read is back, and so on down the slots (each ds_add takes the
    // place of the read just consumed); operations the compiler has in flight from before are older and only make
    // the waits stricter; the compiler does not see this block's LDS operations (as in accumulate_counts).
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds_raw);
    constexpr uint32_t LUT_OFF = (uint32_t)(TILE_BYTES + (size_t)CAPJ * 8 + ((size_t)CAPL + 2) * 2);
    static_assert(LUT_OFF < 65536u, "the table's place as the offset field of ds_read");
    auto pair_group3_asm = [&](uint32_t item, uint32_t x1, uint32_t y1, const uint2 (&w2)[GROUP], uint32_t c) {
        const uint32_t rcell = (item >> MK_ROW_SHIFT) & MK_ROW_MASK;
        const uint32_t row_addr = lds_base + rcell * (ROW_Q * 8u);
        const uint32_t s1 = __builtin_amdgcn_perm(x1, x1, 0u);
        const unsigned long long in0 = __ballot(c > 0u), in1 = __ballot(c > 1u), in2 = __ballot(c > 2u);
        uint32_t a0, a1, a2, l0, l1, l2, t;
        unsigned long long v0, v1, v2, sm, saved;
#define MASKS_P1(A, L, V, WX, WY)                                        \
        "v_and_b32 %[" A "], %[x1], %[" WX "]\n\t"                        \
        "v_and_b32 %[t], %[y1], %[" WY "]\n\t"                            \
        "v_and_b32 %[" L "], %[s1], %[" WY "]\n\t"                        \
        "v_bcnt_u32_b32 %[t], %[t], 0\n\t"                                \
        "v_bcnt_u32_b32 %[" L "], %[" L "], 0\n\t"                        \
        "v_cmp_lt_u32_e32 vcc, 0x0fffffff, %[" A "]\n\t"                  \
        "v_addc_co_u32_e64 %[t], vcc, 0, %[t], vcc\n\t"                   \
        "v_mad_u32_u24 %[" L "], %[t], 11, %[" L "]\n\t"                  \
        "v_lshl_add_u32 %[" L "], %[" L "], 3, %[ldsb]\n\t"               \
        "ds_read_b64 %[" V "], %[" L "] offset:%[lutoff]\n\t"
#define MASKS_P2(A, L, V, WX, IN)                                        \
        "v_and_b32 %[t], 0x05ff0000, %[" A "]\n\t"                        \
        "v_cmp_gt_u32_e32 vcc, 0x01000000, %[t]\n\t"                      \
        "s_and_b64 %[sm], vcc, %[" IN "]\n\t"                             \
        "v_addc_co_u32_e64 %[upd], vcc, 0, %[upd], %[sm]\n\t"             \
        "v_cmp_eq_u32_e32 vcc, 0, %[t]\n\t"                               \
        "s_and_b64 %[sm], vcc, %[" IN "]\n\t"                             \
        "v_addc_co_u32_e64 %[adds], vcc, 0, %[adds], %[sm]\n\t"           \
        "v_add_u32_sdwa %[" L "], %[row], %[" WX "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t" \
        "s_waitcnt lgkmcnt(2)\n\t"                                        \
        "s_and_saveexec_b64 %[saved], %[sm]\n\t"                          \
        "ds_add_u64 %[" L "], %[" V "]\n\t"                               \
        "s_mov_b64 exec, %[saved]\n\t"
        asm volatile(MASKS_P1("a0", "l0", "v0", "wx0", "wy0") MASKS_P1("a1", "l1", "v1", "wx1", "wy1")
                     MASKS_P1("a2", "l2", "v2", "wx2", "wy2")
                     MASKS_P2("a0", "l0", "v0", "wx0", "in0") MASKS_P2("a1", "l1", "v1", "wx1", "in1")
                     MASKS_P2("a2", "l2", "v2", "wx2", "in2")
                     : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [l0] "=&v"(l0), [l1] "=&v"(l1), [l2] "=&v"(l2),
                       [t] "=&v"(t), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [sm] "=&s"(sm), [saved] "=&s"(saved),
                       [upd] "+v"(upd_lane), [adds] "+v"(add_lane)
                     : [x1] "v"(x1), [y1] "v"(y1), [s1] "v"(s1), [row] "v"(row_addr), [ldsb] "s"(lds_base),
                       [wx0] "v"(w2[0].x), [wy0] "v"(w2[0].y), [wx1] "v"(w2[1 % GROUP].x), [wy1] "v"(w2[1 % GROUP].y),
                       [wx2] "v"(w2[2 % GROUP].x), [wy2] "v"(w2[2 % GROUP].y), [in0] "s"(in0), [in1] "s"(in1), [in2] "s"(in2),
                       [lutoff] "n"(LUT_OFF)
                     : "vcc", "memory");
#undef MASKS_P1
#undef MASKS_P2
    };
    auto group_load = [&](uint32_t item, uint2 (&w2)[GROUP]) {
        const uint2 *p = sJ + (item & MK_J_MASK);
        // (lanes with fewer than GROUP read on inside the staging area or the offsets behind it: idle in those slots)
#pragma unroll
        for (int u = 0; u < GROUP; ++u) w2[u] = p[u];
    };
    auto pair_dispatch = [&](uint32_t item, uint32_t x1, uint32_t y1, const uint2 (&w2)[GROUP], uint32_t c) {
        if (DIAG) pair_group(std::true_type{}, item, x1, y1, w2, c);
        else if (GROUP == 3 && slot_asm) pair_group3_asm(item, x1, y1, w2, c);
        else pair_group(std::false_type{}, item, x1, y1, w2, c);
    };
    // `more`: the lanes whose item goes on, `adv` column entries further
    auto ring_push = [&](unsigned long long more, uint32_t item, uint32_t x1, uint32_t y1, uint32_t adv) {
        if (more) {
            if (__builtin_amdgcn_inverse_ballot_w64(more)) {
                const uint32_t slot = (ring_tail + __builtin_amdgcn_mbcnt_hi(
                        (uint32_t)(more >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u))) & (RING - 1);
                ring[slot] = make_uint2(item + adv - (adv << MK_C_SHIFT), x1);
                ring_y[slot] = y1;
            }
            ring_tail += (uint32_t)__popcll(more);
        }
    };
    // An item of the primary batches takes TWO rounds of GROUP column entries in line before what is left of it
    // goes to the ring (the second round's entries are requested before the first is paired): with 5.8 column
    // entries per item on C3 clustered 80 % of the items went through the ring once and 35 % twice, and a trip
    // through the ring costs about as many instructions as two slots.
    auto group_pair = [&](uint32_t item, uint32_t x1, uint32_t y1, const uint2 (&w2)[GROUP]) {
        const uint32_t c = item >> MK_C_SHIFT;
        const unsigned long long more1 = __ballot(c > (uint32_t)GROUP);
        uint2 wd[GROUP];
        if (more1) group_load(item + (uint32_t)GROUP, wd);
        pair_dispatch(item, x1, y1, w2, c);
        if (more1) {
            pair_dispatch(item, x1, y1, wd, c > (uint32_t)GROUP ? c - (uint32_t)GROUP : 0u);
            ring_push(__ballot(c > 2u * GROUP), item, x1, y1, 2u * GROUP);
        }
    };
    // One batch from the ring (up to 64 items). The ring must be empty before the next range is staged, and its
    // last batches are thin -- 50 items, then the 20 of them that go on, then 8, 3, 1: as many passes again as the
    // full ones, at a quarter of the lanes (C3 clustered: 44 % of the lane slots of a launch held a pair). A batch
    // of 32 items or fewer is therefore spread over the wave: 2 / 4 / 8 / 16 lanes per item, lane `sub` of an
    // item taking its column entries from sub * GROUP on -- one pass covers 2 ... 16 x GROUP entries of every
    // item, and only what is beyond that goes round again (pushed by the item's lane 0).
    auto drain_one = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t n = min(64u, ring_tail - ring_head);
        const uint32_t k = n > 32u ? 0u : n > 16u ? 1u : n > 8u ? 2u : n > 4u ? 3u : 4u;  // log2(lanes per item)
        const uint32_t idx = lane & ((64u >> k) - 1u), sub = lane >> (6u - k);
        uint2 it = make_uint2(0u, 0u);
        uint32_t y = 0u;
        if (idx < n) {
            it = ring[(ring_head + idx) & (RING - 1)];
            y = ring_y[(ring_head + idx) & (RING - 1)];
        }
        __builtin_amdgcn_wave_barrier();
        ring_head += n;
        const uint32_t c0 = it.x >> MK_C_SHIFT, adj = sub * (uint32_t)GROUP;
        const uint32_t cl = c0 > adj ? c0 - adj : 0u;
        const uint32_t item_l = ((it.x & ~(0xFFu << MK_C_SHIFT)) + adj) | (cl << MK_C_SHIFT);
        uint2 w2[GROUP];
        group_load(item_l, w2);
        const uint32_t adv = (uint32_t)GROUP << k;
        const unsigned long long more = __ballot(sub == 0u && c0 > adv);
        pair_dispatch(item_l, it.y, y, w2, cl);
        ring_push(more, it.x, it.y, y, adv);
    };

    for (uint32_t r = r_begin; r < r_end; ++r) {
        const uint32_t la = n_la, ib = n_ib, ie = n_ie, jb = n_jb, dsh = n_dsh;
        const bool staged = n_staged;
        __syncthreads();  // every wave is done with the previous range (first time: with zeroing)
        if (staged) {
            static_assert(JPT * THREADS <= CAPJ && OPT * THREADS <= CAPL + 2, "unconditional staging stores");
#pragma unroll
            for (int k = 0; k < JPT; ++k)
                sJ[tid + k * THREADS] = make_uint2(pJ[k], pMj[k]);
#pragma unroll
            for (int k = 0; k < OPT; ++k) sOff[tid + k * THREADS] = (uint16_t)(pO[k] - jb);
        }
        __syncthreads();
        const uint32_t nI = ie - ib;
        upd_lane = 0;
        add_lane = 0;
        if (staged) {
            uint32_t item[JPT], ix[JPT], iy[JPT];
            uint32_t any_wide = 0;
#pragma unroll
            for (int k = 0; k < JPT; ++k) {
                const uint32_t i = tid + k * THREADS;
                const uint32_t rec = pI[k];
                const uint32_t lrel = rec >> 16;
                uint32_t j0 = sOff[lrel];
                const uint32_t j1 = sOff[lrel + 1];
                if (DIAG) j0 = i + dsh + 1u;  // a diagonal tile: the entries after this one, each pair once
                // (a row entry whose read reaches beyond its windows pairs in wide_pairs only)
                const uint32_t c = (i < nI && j1 > j0 && (rec & C_WIDE) == 0u) ? j1 - j0 : 0u;
                any_wide |= c;
                item[k] = (j0 & MK_J_MASK) | ((rec & C_CELL) << MK_ROW_SHIFT) | (c << MK_C_SHIFT);
                ix[k] = pXi[k];
                iy[k] = pMi[k];
            }
            // wide items (32 column entries or more): the whole wave pairs one row entry with 64 at a time
            if (__ballot(any_wide >= IT_WIDE)) {
#pragma unroll
                for (int k = 0; k < JPT; ++k) {
                    // (the count again, in full: an item holds 8 bits of it)
                    const uint32_t i = tid + k * THREADS;
                    const uint32_t rec = pI[k];
                    const uint32_t lrel = rec >> 16;
                    uint32_t j0 = sOff[lrel];
                    const uint32_t j1 = sOff[lrel + 1];
                    if (DIAG) j0 = i + dsh + 1u;
                    const uint32_t c = (i < nI && j1 > j0 && (rec & C_WIDE) == 0u) ? j1 - j0 : 0u;
                    unsigned long long todo = __ballot(c >= IT_WIDE);
                    while (todo) {
                        const int src = __builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        const uint32_t itw = __builtin_amdgcn_readlane(item[k], src);
                        const uint32_t xw = __builtin_amdgcn_readlane(ix[k], src);
                        const uint32_t yw = __builtin_amdgcn_readlane(iy[k], src);
                        const uint32_t cw = __builtin_amdgcn_readlane(c, src);
                        const uint32_t j0w = __builtin_amdgcn_readlane(j0, src);
                        for (uint32_t base = 0; base < cw; base += 64u) {
                            uint2 w2[GROUP];
                            w2[0] = sJ[min(j0w + base + lane, (uint32_t)CAPJ - 1u)];
#pragma unroll
                            for (int u = 1; u < GROUP; ++u) w2[u] = w2[0];
                            const uint32_t c1 = base + lane < cw ? 1u : 0u;  // slot 0 only
                            if (DIAG) pair_group(std::true_type{}, itw, xw, yw, w2, c1);
                            else pair_group(std::false_type{}, itw, xw, yw, w2, c1);
                        }
                    }
                    if (c >= IT_WIDE) item[k] = 0u;  // done
                }
            }
            if (r + 1 < r_end) prefetch(r + 1);  // the column-side registers are free again
            {
                uint2 wc[GROUP], wn[GROUP];
                group_load(item[0], wc);
#pragma unroll
                for (int k = 0; k < JPT; ++k) {
                    if (k + 1 < JPT) group_load(item[k + 1], wn);
                    while (ring_tail - ring_head > (uint32_t)(RING - 64)) drain_one();  // room for 64 continuations
                    group_pair(item[k], ix[k], iy[k], wc);
#pragma unroll
                    for (int u = 0; u < GROUP; ++u) wc[u] = wn[u];
                }
            }
            if (r + 1 < r_end) prefetch_rows();
            while (ring_tail != ring_head) drain_one();
        } else {
            if (r + 1 < r_end) {
                prefetch(r + 1);
                prefetch_rows();
            }
            // a locus range that does not fit the staging buffers (a single very deep locus): paired straight
            // from HBM/L2 with the compact entries, straight into HBM (as accumulate_tiles does)
            const long long d10 = a.lut[1 * LUT_DIM + 0], d01 = a.lut[0 * LUT_DIM + 1];
            uint32_t upd = 0, skipped = 0;
            for (uint32_t e1 = ib + tid; e1 < ie; e1 += THREADS) {
                const uint32_t r1 = a.entry32[e1];
                const uint32_t l = la + (r1 >> 16);
                const uint32_t j0 = DIAG ? e1 + 1 : offJ[l];
                const uint32_t j1 = offJ[l + 1];
                const uint32_t row = (r1 & C_CELL) * B;
                for (uint32_t e2 = j0; e2 < j1; ++e2) {
                    const uint32_t r2 = a.entry32[e2];
                    const uint32_t x = r1 ^ r2, both = r1 & r2;
                    if (DIAG && (x & C_CELL) == 0u) continue;  // same cell (:215)
                    if (both & C_TAIL) continue;               // both never flushed (:407-408)
                    if ((r1 | r2) & C_WIDE) continue;          // wide_pairs
                    ++upd;
                    long long v = (x & (3u << C_BASE_SHIFT)) ? d01 : d10;
                    if (both & C_MULTI) {
                        v = pair_value_full(a.slow, e1, e2);
                        if (v == NO_PAIR) {
                            ++skipped;
                            continue;
                        }
                    }
                    atomicAdd(&dst[row + (r2 & C_CELL)], (unsigned long long)v);
                }
            }
            n_updates += upd;
            n_pairs += (unsigned long long)upd - skipped;
        }
        n_updates += upd_lane;
        n_pairs += add_lane;
    }
    __syncthreads();

    // flush: the tile goes to the workgroup's own slab with plain coalesced stores; reduce_slabs adds up
    {
        unsigned long long *out = reinterpret_cast<unsigned long long *>(a.slab) + (size_t)blockIdx.x * B * B;
        for (uint32_t i = tid; i < B * B; i += THREADS) out[i] = tile64[(i / B) * ROW_Q + (i % B)];  // dense rows
    }
    // work counters: wave reduction, then one atomic pair per workgroup (see accumulate_tiles)
    for (int off = 32; off > 0; off >>= 1) {
        n_updates += __shfl_down(n_updates, off);
        n_pairs += __shfl_down(n_pairs, off);
    }
    unsigned long long *red = reinterpret_cast<unsigned long long *>(sJ);
    if (lane == 0u) {
        red[(tid >> 6) * 2] = n_updates;
        red[(tid >> 6) * 2 + 1] = n_pairs;
    }
    __syncthreads();
    if (tid == 0u) {
        unsigned long long u = 0, q = 0;
        for (int w = 0; w < WAVES; ++w) {
            u += red[w * 2];
            q += red[w * 2 + 1];
        }
        if (u | q) {
            atomicAdd(&a.counters[0], u);
            atomicAdd(&a.counters[1], q);
        }
    }
}

// The two words of every packed entry as accumulate_masks pairs from them, made ONCE per prepare (a pass over the
// packed entries: 8 bytes read, 12 written) instead of by every workgroup for every entry of its row and of its column
// block -- 125 times each on C3 clustered, a tenth of the kernel's vector instructions: y = the base planes, x in
// the column form (low half: the cell's byte offset in a tile row; dead = the read reaches beyond its windows) and
// in the row form (low half: next8; dead set).
__global__ __launch_bounds__(256) void k_masks_words(const uint32_t *entry32, const uint32_t *mask32, uint32_t n,
                                                    uint32_t *y, uint32_t *xcol, uint32_t *xrow) {
    for (uint32_t d = blockIdx.x * 256 + threadIdx.x; d < n; d += gridDim.x * 256) {
        const uint32_t e = entry32[d], m = mask32[d];
        const uint32_t f = masks_flags(e, m);
        y[d] = masks_planes(m);
        xcol[d] = f | ((e & C_CELL) << 3);
        xrow[d] = f | MX_DEAD | ((m >> 8) & 0xFFu);
    }
}

// ---- reads that reach beyond their 8-locus windows (C_WIDE): rare where loci are a read length apart (C2 clustered:
// 0.01 % of the reads), and what they need -- the 16-byte records, the entry indices, sometimes the merge walk --
// is what the hot loop of accumulate_masks does without. Their entries are listed per cell block once per prepare
// (count, scan, fill: entries of a block are contiguous in the packed arrays), and wide_pairs adds every pair
// with at least one of them to the accumulator after the main kernel: a workgroup per tile, a thread per
// (wide entry, partner entry) run.
__global__ __launch_bounds__(256) void k_wide_count(const uint32_t *entry32, const uint32_t *blk_off, uint32_t stride,
                                                   uint32_t *cnt) {
    const uint32_t b = blockIdx.y, e0 = blk_off[(size_t)b * stride], e1 = blk_off[(size_t)b * stride + stride - 1u];
    uint32_t n = 0;
    for (uint32_t d = e0 + blockIdx.x * 256 + threadIdx.x; d < e1; d += gridDim.x * 256) n += (entry32[d] & C_WIDE) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if ((threadIdx.x & 63u) == 0u && n) atomicAdd(&cnt[b], n);
}
// off[0 .. nb] = exclusive scan of cnt[0 .. nb), cur[b] = off[b] (the fill's cursors); one workgroup
__global__ __launch_bounds__(1024) void k_wide_scan(const uint32_t *cnt, uint32_t nb, uint32_t *off, uint32_t *cur) {
    __shared__ uint32_t s[1024];
    const uint32_t tid = threadIdx.x;
    s[tid] = tid < nb ? cnt[tid] : 0u;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? s[tid - d] : 0u;
        __syncthreads();
        s[tid] += v;
        __syncthreads();
    }
    if (tid < nb) {
        const uint32_t ex = s[tid] - cnt[tid];
        off[tid] = ex;
        cur[tid] = ex;
    }
    if (tid == nb - 1u) off[nb] = s[tid];
}
__global__ __launch_bounds__(256) void k_wide_fill(const uint32_t *entry32, const uint32_t *blk_off, uint32_t stride,
                                                  uint32_t *cur, uint32_t *list) {
    const uint32_t b = blockIdx.y, e0 = blk_off[(size_t)b * stride], e1 = blk_off[(size_t)b * stride + stride - 1u];
    for (uint32_t d = e0 + blockIdx.x * 256 + threadIdx.x; d < e1; d += gridDim.x * 256)
        if (entry32[d] & C_WIDE) list[atomicAdd(&cur[b], 1u)] = d;
}

struct WideArgs {
    const uint32_t *blk_off;
    uint32_t stride;
    const uint32_t *entry32;
    const uint4 *entry;
    const SlowPathArgs *slow;
    const long long *lut;
    const uint16_t *tile_row, *tile_col;
    uint32_t tile_begin;
    const uint32_t *tile_ids;
    const uint32_t *wide_off, *wide_list;
    long long *acc;
    unsigned long long *counters;
};
template <int B>
__global__ __launch_bounds__(256) void wide_pairs(const WideArgs a) {
    const uint32_t t = a.tile_ids ? a.tile_ids[blockIdx.x] : a.tile_begin + blockIdx.x;
    const uint32_t I = a.tile_row[t], J = a.tile_col[t];
    const bool diag = I == J;
    const long long d10 = a.lut[1 * LUT_DIM + 0], d01 = a.lut[0 * LUT_DIM + 1];
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.acc) + (size_t)t * B * B;
    uint32_t upd = 0, skipped = 0;
    // side 0: the wide entries of block I as row entries against every entry of block J at their locus (in a
    // diagonal tile: against the non-wide entries, and the wide ones behind them -- each pair once);
    // side 1 (I != J): the wide entries of block J as column entries against the NON-wide entries of block I
    for (int side = 0; side < (diag ? 1 : 2); ++side) {
        const uint32_t bw = side ? J : I, bo = side ? I : J;
        const uint32_t *off_o = a.blk_off + (size_t)bo * a.stride;
        // (a WAVE per wide entry, its lanes over the other block's entries of the locus: a thread per wide entry walked
        // them one after the other, each with the slow path's chain of gathers behind it -- 91 us for the 22 wide
        // entries per block of C2 clustered)
        for (uint32_t k = a.wide_off[bw] + (threadIdx.x >> 6); k < a.wide_off[bw + 1]; k += 4) {
            const uint32_t ew = a.wide_list[k];
            const uint32_t rw = a.entry32[ew];
            const uint32_t l = a.entry[ew].w;  // (every wide entry is multi-locus: it has its 16-byte record)
            for (uint32_t eo = off_o[l] + (threadIdx.x & 63u); eo < off_o[l + 1]; eo += 64u) {
                const uint32_t ro = a.entry32[eo];
                if (ro & C_WIDE) {
                    if (side == 1) continue;          // wide x wide: side 0 has it
                    if (diag && eo <= ew) continue;   // ... and in a diagonal tile the earlier of the two
                } else if (diag && eo == ew) {
                    continue;
                }
                const uint32_t x = rw ^ ro, both = rw & ro;
                if (diag && (x & C_CELL) == 0u) continue;  // same cell (:215)
                if (both & C_TAIL) continue;               // both never flushed (:407-408)
                ++upd;
                long long v = (x & (3u << C_BASE_SHIFT)) ? d01 : d10;
                const uint32_t e_row = side ? eo : ew, e_col = side ? ew : eo;
                if (both & C_MULTI) {
                    v = pair_value_full(a.slow, e_row, e_col);
                    if (v == NO_PAIR) {
                        ++skipped;
                        continue;
                    }
                }
                const uint32_t r_row = side ? ro : rw, r_col = side ? rw : ro;
                atomicAdd(&dst[(r_row & C_CELL) * B + (r_col & C_CELL)], (unsigned long long)v);
            }
        }
    }
    unsigned long long u = upd, q = (unsigned long long)upd - skipped;
    for (int off = 32; off > 0; off >>= 1) {
        u += __shfl_down(u, off);
        q += __shfl_down(q, off);
    }
    if ((threadIdx.x & 63u) == 0u && (u | q)) {
        atomicAdd(&a.counters[0], u);
        atomicAdd(&a.counters[1], q);
    }
}

// ---- the flagged entries (tail: the read was never flushed; multi: the read has further kept entries), compacted
// in the order of the packed entries, i.e. by (cell block, locus): pre[d] = flagged entries before entry d, so
// the flagged entries of the group (b, l) are [pre[blk_off[b][l]], pre[blk_off[b][l + 1]]) of the compact list
struct FlaggedOp {  // input of the prefix sum
    const uint32_t *entry32;
    uint32_t n;
    __device__ __forceinline__ uint32_t operator()(uint32_t d) const {
        return (d < n && (entry32[d] & (C_TAIL | C_MULTI)) != 0u) ? 1u : 0u;
    }
};
__global__ __launch_bounds__(256) void flagged_compact(const uint32_t *entry32, const uint4 *entry, uint32_t n,
                                                      const uint32_t *pre, uint4 *rec, uint32_t *idx) {
    for (uint32_t d = blockIdx.x * 256 + threadIdx.x; d < n; d += gridDim.x * 256)
        if (entry32[d] & (C_TAIL | C_MULTI)) {
            const uint32_t i = pre[d];
            rec[i] = entry[d];
            idx[i] = d;
        }
}

// ... and per (block, locus) group the number of flagged entries before it
__global__ __launch_bounds__(256) void flagged_groups(const uint32_t *blk_off, size_t n_off, const uint32_t *pre,
                                                     uint32_t *grp) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_off; i += (size_t)gridDim.x * 256)
        grp[i] = pre[blk_off[i]];
}

// x_s, x_d over all loci two multi-locus reads share, packed (values, not out-parameters: those would live
// in scratch memory around the out-of-line call): kJointOwned if the locus of both entries is the first one
// they share (else an earlier locus owns the pair) | x_s << 15 | x_d
constexpr uint32_t kJointOwned = 1u << 31;
__device__ __forceinline__ uint32_t joint_pack(bool owned, uint32_t xs, uint32_t xd) {
    return (owned ? kJointOwned : 0u) | (xs << 15) | xd;
}
// a 16-locus window overflowed on the same side for both reads: merge-walk their entry lists (rare: out of line)
__device__ __noinline__ uint32_t joint_counts_walk(const SlowPathArgs *sp, const uint32_t *flag_idx, uint32_t locus,
                                                   uint32_t i1, uint32_t i2) {
    const uint32_t r1 = sp->entry_read[flag_idx[i1]], r2 = sp->entry_read[flag_idx[i2]];
    uint32_t j1 = sp->read_off[r1], e1 = sp->read_off[r1 + 1];
    uint32_t j2 = sp->read_off[r2], e2 = sp->read_off[r2 + 1];
    uint32_t xs = 0, xd = 0, first = 0xFFFFFFFFu;
    while (j1 < e1 && j2 < e2) {
        const uint32_t l1 = sp->read_locus[j1], l2 = sp->read_locus[j2];
        if (l1 == l2) {
            if (first == 0xFFFFFFFFu) first = l1;
            if (sp->read_base[j1] == sp->read_base[j2]) ++xs; else ++xd;
            ++j1; ++j2;
        } else if (l1 < l2) {
            ++j1;
        } else {
            ++j2;
        }
    }
    return joint_pack(first == locus, xs, xd);
}

// The logic of pair_value_full / slow_pair. i1, i2: the entries' places in the compact list (their entry
// indices are looked up only on the slow path).
__device__ __forceinline__ uint32_t joint_counts(const SlowPathArgs *sp, const uint32_t *flag_idx, const uint4 A1,
                                                 const uint4 A2, uint32_t i1, uint32_t i2) {
    if (A1.y & A2.y & 0xFFFFu) return 0u;  // they share an earlier locus
    const bool same = (((A1.x ^ A2.x) >> 16) & 3u) == 0u;
    if ((A1.x & A2.x & (META_PREV_OVF | META_NEXT_OVF)) == 0u) {
        const uint32_t shared = (A1.y & A2.y) >> 16;
        const uint32_t x = A1.z ^ A2.z;
        const uint32_t diff = ((x & 0xFFFFu) | (x >> 16)) & shared;
        const uint32_t nd = __popc(diff);
        return joint_pack(true, __popc(shared) - nd + (same ? 1u : 0u), nd + (same ? 0u : 1u));
    }
    return joint_counts_walk(sp, flag_idx, A1.w, i1, i2);
}

struct CorrectArgs {
    const uint32_t *blk_off;   // num_blocks * stride (the packed pileup's group offsets)
    uint32_t stride;           // num_loci + 1
    const uint32_t *flag_grp;  // num_blocks * stride: flagged entries before each (block, locus) group
    const uint4 *flag_rec;     // the flagged entries' records, compact
    const uint32_t *flag_idx;  // ... and entry indices
    const SlowPathArgs *slow;
    const long long *lut;
    const uint16_t *tile_row, *tile_col;
    uint32_t tile_begin;       // tiles of this launch: [tile_begin, tile_begin + gridDim.x) ...
    const uint32_t *tile_ids;  // ... or, when non-null, by global index
    const void *slab;          // accumulate_counts' count tiles, one per workgroup of its launch
    const uint32_t *tile_wg_begin;
    int64_t *acc;
    unsigned long long *counters;
    uint32_t split;            // workgroups per tile
    uint32_t overwrite;        // acc[tile] = result instead of += (split == 1; the launcher zeroes otherwise)
    unsigned long long *max_bits;  // when non-null (split == 1): atomicMax of max(0, max D) over the tile, as
    double max_scale;              // reduce_max would find it in the stored tile (D = value * max_scale)
};

// One workgroup per tile (I, J) of the launch, after accumulate_counts. Two things in one pass over the tile:
//
// 1. What the flags mean. Every pair of flagged entries of one locus, one in cell block I and one in J, of
//    two different matrix rows:
//      both tail            -> the plain term accumulate_counts added is taken out, so is its update
//      both multi-locus     -> not the first shared locus: one read pair less (the incidence stays an update);
//                              first shared locus and n = x_s + x_d >= 2: + D(x_s,x_d) - x_s D(1,0) - x_d D(0,1)
//    A thread takes a flagged entry p of block I and walks the flagged entries q of the same locus in block J
//    (in a diagonal tile those behind p: every pair once); the terms are added into an int64 tile in LDS.
//    (Round 2 first did this per locus with one scattered 8-byte global atomic per term -- 23 million of them
//    on C3, at the ~2e10 per second the memory side serves chip-wide: 1.2 ms, and beside the pair kernel it
//    cost that kernel 0.8 ms. In LDS the same terms are a few tens of microseconds per tile.)
// 2. reduce_slabs: the count tiles of the tile's accumulate_counts workgroups, converted with the two
//    single-locus ratios (exact integer arithmetic), plus the LDS tile, added to acc[tile] with plain coalesced
//    read-modify-writes: nothing else touches the tile during the launch.
template <int B, int THREADS>
__global__ __launch_bounds__(THREADS) void correct_tiles(const CorrectArgs a) {
    extern __shared__ unsigned long long corr[];  // B * B
    __shared__ long long part[2 * (THREADS / 64)];
    // the correction D(x_s,x_d) - x_s D(1,0) - x_d D(0,1) itself for few shared loci (the usual case)
    __shared__ long long scorr[SLUT_DIM * SLUT_DIM];
    constexpr int U = 6;  // flagged entries p per thread in flight: their loads are issued together
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t S = a.split, t_local = blockIdx.x / S, part_id = blockIdx.x % S;
    const uint32_t t = a.tile_ids ? a.tile_ids[t_local] : a.tile_begin + t_local;
    const uint32_t I = a.tile_row[t], J = a.tile_col[t];
    const bool diag = I == J;
    for (uint32_t i = tid; i < (uint32_t)(B * B); i += THREADS) corr[i] = 0ull;
    const long long d10 = a.lut[1 * LUT_DIM + 0], d01 = a.lut[0 * LUT_DIM + 1];
    if (tid < (uint32_t)(SLUT_DIM * SLUT_DIM)) {
        const uint32_t xs = tid / SLUT_DIM, xd = tid % SLUT_DIM;
        scorr[tid] = a.lut[xs * LUT_DIM + xd] - (long long)xs * d10 - (long long)xd * d01;
    }
    __syncthreads();
    // The usual launch -- one count tile per matrix tile, one workgroup per tile -- asks for its count tile NOW: the
    // 16 words per thread land while the flagged entries are paired (the flush below then starts with its data at
    // hand instead of with a round trip to HBM).
    constexpr int PRE = (B * B) / THREADS;
    const bool prefetched = S == 1u && a.tile_wg_begin[t_local + 1] - a.tile_wg_begin[t_local] == 1u && PRE <= 16;
    uint32_t pre[PRE <= 16 ? PRE : 1];
    if (prefetched) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.slab) + (size_t)a.tile_wg_begin[t_local] * B * B;
#pragma unroll
        for (int k = 0; k < (PRE <= 16 ? PRE : 1); ++k) pre[k] = __builtin_nontemporal_load(&src[tid + (uint32_t)k * THREADS]);
    }
    long long upd_delta = 0, pair_delta = 0;  // per lane
#ifdef SECEDO_STAMPS
    unsigned long long dg_tests = 0, dg_tail = 0, dg_joint = 0, dg_later = 0;
    long long dg_loads = 0, dg_qwait = 0, dg_qiter = 0;
    uint32_t dg_sink = 0;
    const long long dg_t0 = __builtin_readcyclecounter();
#endif
    const size_t rowI = (size_t)I * a.stride, rowJ = (size_t)J * a.stride;
    const uint32_t p0 = a.flag_grp[rowI], p1 = a.flag_grp[rowI + a.stride - 1u];
    for (uint32_t base = p0 + part_id * (uint32_t)(THREADS * U); base < p1; base += S * (uint32_t)(THREADS * U)) {
        uint4 A1[U];
        uint32_t qa[U], qb[U];
#ifdef SECEDO_STAMPS
        const long long dg_b0 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = base + (uint32_t)u * THREADS + tid;
            A1[u] = p < p1 ? a.flag_rec[p] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = base + (uint32_t)u * THREADS + tid;
            const size_t g = rowJ + A1[u].w;
            qa[u] = p < p1 ? a.flag_grp[g] : 0u;
            qb[u] = p < p1 ? a.flag_grp[g + 1] : 0u;
            if (diag) qa[u] = p + 1u;  // p lies in its own group: every pair once
        }
        // (six entries in flight and no prefetch of the first q record: 103 VGPRs; with the prefetch four fit,
        // and eight spill. Advancing the q of all entries together, U loads per round trip, was no faster on C3
        // and slower on C2: the rounds are set by the longest list among the lanes either way.)
#ifdef SECEDO_STAMPS
        dg_sink += qa[0] + qb[U - 1];  // (the loads have arrived)
        const long long dg_b1 = __builtin_readcyclecounter();
        dg_loads += dg_b1 - dg_b0;
#endif
        // One flagged column entry q against the flagged row entry p (P its record, Q the column entry's).
        auto pair_pq = [&](const uint4 P, uint32_t p, const uint4 Q, uint32_t q) {
            const uint32_t row1 = (P.x & 0xFFFFu) - I * B, row2 = (Q.x & 0xFFFFu) - J * B;
            if (diag && row1 == row2) return;  // same cell (:215)
#ifdef SECEDO_STAMPS
            ++dg_tests;
#endif
            const bool multi1 = P.y != 0u || (P.x & (META_PREV_OVF | META_NEXT_OVF)) != 0u;
            const bool tails = (P.x & Q.x & (1u << 18)) != 0u;  // both never flushed: no pair at all
            const bool multi2 = Q.y != 0u || (Q.x & (META_PREV_OVF | META_NEXT_OVF)) != 0u;
            if (!tails && !(multi1 && multi2)) return;
            // (inside a diagonal tile either orientation is read back: the finalize kernels add both)
            unsigned long long *cell = &corr[row1 * B + row2];
            const bool same = (((P.x ^ Q.x) >> 16) & 3u) == 0u;
            if (tails) {
                atomicAdd(cell, (unsigned long long)(-(same ? d10 : d01)));
                --upd_delta;
                --pair_delta;
#ifdef SECEDO_STAMPS
                ++dg_tail;
#endif
                return;
            }
            // (the order of the two entries does not matter to joint_counts)
            const uint32_t jc = joint_counts(a.slow, a.flag_idx, P, Q, p, q);
            const uint32_t xs = (jc >> 15) & 0xFFFFu, xd = jc & 0x7FFFu;
            if ((jc & kJointOwned) == 0u) {
                --pair_delta;  // counted at their first shared locus
#ifdef SECEDO_STAMPS
                ++dg_later;
#endif
                return;
            }
            if (xs + xd < 2u) return;  // this locus only: the plain term is the whole term
            long long term;
            if (xs < (uint32_t)SLUT_DIM && xd < (uint32_t)SLUT_DIM) {
                term = scorr[xs * SLUT_DIM + xd];
            } else {
                const long long joint = xs + xd <= REF_TABLE_MAX
                        ? a.lut[xs * LUT_DIM + xd] : llr_fixed_device(a.slow, xs, xd, P.x & 0xFFFFu, Q.x & 0xFFFFu);
                term = joint - (long long)xs * d10 - (long long)xd * d01;
            }
            atomicAdd(cell, (unsigned long long)term);
#ifdef SECEDO_STAMPS
            ++dg_joint;
#endif
        };
        // The column lists of TWO row entries advance together: a wave walks a list as long as its longest lane
        // needs, one memory round trip per step, so six lists one after the other cost the sum of six maxima (48
        // steps on C3 where the average list has 2.8 entries); in pairs it is three maxima of two, with two records
        // in flight per lane. (All six together -- 24 more registers -- was measured no faster in round 2.)
        static_assert(U % 2 == 0, "lists in pairs");
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            const uint32_t pA = base + (uint32_t)u * THREADS + tid, pB = pA + THREADS;
            uint32_t qA = qa[u], qB = qa[u + 1];
            const uint32_t eA = qb[u], eB = qb[u + 1];
            while (qA < eA || qB < eB) {
                const bool hA = qA < eA, hB = qB < eB;
                uint4 QA = make_uint4(0, 0, 0, 0), QB = make_uint4(0, 0, 0, 0);
                if (hA) QA = a.flag_rec[qA];
                if (hB) QB = a.flag_rec[qB];
#ifdef SECEDO_STAMPS
                ++dg_qiter;
#endif
                if (hA) pair_pq(A1[u], pA, QA, qA);
                if (hB) pair_pq(A1[u + 1], pB, QB, qB);
                qA += hA ? 1u : 0u;
                qB += hB ? 1u : 0u;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        upd_delta += __shfl_down(upd_delta, off);
        pair_delta += __shfl_down(pair_delta, off);
    }
    if (lane == 0u) {
        part[(tid >> 6) * 2] = upd_delta;
        part[(tid >> 6) * 2 + 1] = pair_delta;
    }
#ifdef SECEDO_STAMPS
    for (int off = 32; off > 0; off >>= 1) {
        dg_tests += __shfl_down(dg_tests, off);
        dg_tail += __shfl_down(dg_tail, off);
        dg_joint += __shfl_down(dg_joint, off);
        dg_later += __shfl_down(dg_later, off);
    }
    for (int off = 32; off > 0; off >>= 1) {  // the busiest lane stands for the wave
        dg_qwait = max(dg_qwait, (long long)__shfl_down(dg_qwait, off));
        dg_qiter = max(dg_qiter, (long long)__shfl_down(dg_qiter, off));
    }
    __shared__ unsigned long long dg_part[4];
    if (tid < 4u) dg_part[tid] = 0ull;
    __syncthreads();
    if (lane == 0u) {
        atomicAdd(&dg_part[0], dg_tests);
        atomicAdd(&dg_part[1], dg_tail);
        atomicAdd(&dg_part[2], dg_joint);
        atomicAdd(&dg_part[3], dg_later);
    }
    const long long dg_t1 = __builtin_readcyclecounter();
#endif
    __syncthreads();
    if (tid == 0u) {
        long long u = 0, q = 0;
        for (int w = 0; w < THREADS / 64; ++w) {
            u += part[2 * w];
            q += part[2 * w + 1];
        }
        if (u) atomicAdd(&a.counters[0], (unsigned long long)u);
        if (q) atomicAdd(&a.counters[1], (unsigned long long)q);
    }
    // the count tiles of the pair kernel's workgroups of this tile + the corrections -> acc. (A tile shared
    // by S workgroups -- launches of few tiles -- adds with atomics, each workgroup its own corrections and
    // every S-th row of the count tiles.)
    const uint32_t w0 = a.tile_wg_begin[t_local], w1 = a.tile_wg_begin[t_local + 1];
    long long *dst = reinterpret_cast<long long *>(a.acc) + (size_t)t * B * B;
    // (a thread has F cells in flight: their loads are issued together, the tile is two round trips to HBM)
    long long best = 0;  // max(0, max D) over the tile, in accumulator units
    constexpr int F = (B * B / THREADS) < 8 ? (B * B / THREADS) : 8;
    static_assert((B * B) % (THREADS * F) == 0, "whole batches");
    const uint32_t *slab32 = reinterpret_cast<const uint32_t *>(a.slab);
    int batch = 0;
    for (uint32_t c0 = tid; c0 < (uint32_t)(B * B); c0 += THREADS * F, ++batch) {
        uint32_t n_same[F], n_diff[F];
        long long old[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            n_same[f] = 0;
            n_diff[f] = 0;
            old[f] = (S == 1u && !a.overwrite) ? dst[c0 + f * THREADS] : 0ll;
        }
        if (prefetched) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                // (cell c0 + f * THREADS = tid + (batch * F + f) * THREADS: the word pre[batch * F + f])
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < (PRE <= 16 ? PRE : 1); ++k) v = (k == batch * F + f) ? pre[k] : v;
                n_same[f] = v & 0xFFFFu;
                n_diff[f] = v >> 16;
            }
        }
        for (uint32_t w = prefetched ? w1 : w0; w < w1; ++w) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const uint32_t c = c0 + f * THREADS;
                if (S == 1u || (c / THREADS) % S == part_id) {
                    const uint32_t v = slab32[(size_t)w * B * B + c];
                    n_same[f] += v & 0xFFFFu;
                    n_diff[f] += v >> 16;
                }
            }
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const uint32_t c = c0 + f * THREADS;
            const long long sum = (long long)corr[c] + (long long)n_same[f] * d10 + (long long)n_diff[f] * d01;
            if (S == 1u) {
                if (sum || a.overwrite) dst[c] = old[f] + sum;
                if (a.max_bits) {
                    if (diag) corr[c] = (unsigned long long)(old[f] + sum);  // (paired with its mirror below)
                    else best = max(best, old[f] + sum);
                }
            } else if (sum) {
                atomicAdd(reinterpret_cast<unsigned long long *>(&dst[c]), (unsigned long long)sum);
            }
        }
    }
    if (a.max_bits) {  // (wave-uniform) the maximum finalize needs, while the tile is at hand
        if (diag) {    // a diagonal tile holds each pair in either orientation
            __syncthreads();
            for (uint32_t c = tid; c < (uint32_t)(B * B); c += THREADS) {
                const uint32_t r = c / B, col = c % B;
                if (r < col) best = max(best, (long long)corr[c] + (long long)corr[col * B + r]);
            }
        }
        for (int off = 32; off > 0; off >>= 1) best = max(best, (long long)__shfl_down(best, off));
        __syncthreads();  // (part[] was read above)
        if (lane == 0u) part[tid >> 6] = best;
        __syncthreads();
        if (tid == 0u) {
            for (int w = 1; w < THREADS / 64; ++w) best = max(best, part[w]);
            // (double)int64 * 2^-k is monotone, so the max of the integers gives the max of the doubles
            atomicMax(a.max_bits, (unsigned long long)__double_as_longlong((double)best * a.max_scale));
        }
    }
#ifdef SECEDO_STAMPS
    if (tid == 0u) {  // (one wave per workgroup is timed)
        atomicAdd(&a.counters[82], dg_part[0]);
        atomicAdd(&a.counters[83], dg_part[1]);
        atomicAdd(&a.counters[84], dg_part[2]);
        atomicAdd(&a.counters[85], dg_part[3]);
        atomicAdd(&a.counters[86], (unsigned long long)(dg_t1 - dg_t0));
        atomicAdd(&a.counters[87], (unsigned long long)(__builtin_readcyclecounter() - dg_t1));
        atomicAdd(&a.counters[88], 1ull);
        atomicAdd(&a.counters[89], (unsigned long long)(p1 - p0));
        atomicAdd(&a.counters[90], (unsigned long long)dg_loads + (dg_sink == 0xFFFFFFFFu ? 1ull : 0ull));
        atomicAdd(&a.counters[91], (unsigned long long)dg_qwait);
        atomicAdd(&a.counters[92], (unsigned long long)dg_qiter);
    }
#endif
}

// acc[tile] += sum over the tile's workgroups of their slab (count slabs are converted with the two
// single-locus ratios: exact integer arithmetic). One thread per cell pair of a tile.
template <int B, bool COUNTS>
__global__ __launch_bounds__(256) void reduce_slabs(const void *slab, const uint32_t *tile_wg_begin,
                                                   uint32_t tile_begin, const uint32_t *tile_ids,
                                                   const long long *lut, long long *acc) {
    // (launches that are to overwrite the accumulator zero their tiles first: the pair kernels of these
    // variants add to it directly as well, launch_accumulate)
    const uint32_t t_local = blockIdx.x / (B * B / 256);
    const uint32_t cell = (blockIdx.x % (B * B / 256)) * 256 + threadIdx.x;
    const uint32_t w0 = tile_wg_begin[t_local], w1 = tile_wg_begin[t_local + 1];
    long long sum = 0;
    if (COUNTS) {
        const long long d10 = lut[1 * LUT_DIM + 0], d01 = lut[0 * LUT_DIM + 1];
        const uint32_t *p = reinterpret_cast<const uint32_t *>(slab) + cell;
        uint32_t same = 0, diff = 0;
        for (uint32_t w = w0; w < w1; ++w) {
            const uint32_t v = p[(size_t)w * B * B];
            same += v & 0xFFFFu;
            diff += v >> 16;
        }
        sum = (long long)same * d10 + (long long)diff * d01;
    } else {
        const long long *p = reinterpret_cast<const long long *>(slab) + cell;
        for (uint32_t w = w0; w < w1; ++w) sum += p[(size_t)w * B * B];
    }
    long long *dst = &acc[(size_t)(tile_ids ? tile_ids[t_local] : tile_begin + t_local) * B * B + cell];
    if (sum) *dst += sum;
}

// max over i < j of D[i][j], clamped at 0 (the diagonal is zero): bits of a non-negative double
// order like unsigned integers, so atomicMax on the bit pattern is exact. Walks the tile-major
// accumulator linearly (cells beyond num_cells hold zero and do not move a maximum clamped at 0).
template <int B>
__global__ __launch_bounds__(256) void reduce_max(const long long *acc, const uint16_t *tile_row,
                                                  const uint16_t *tile_col, const uint32_t *tile_ids,
                                                  uint32_t n_tiles, double scale, unsigned long long *out_bits) {
    const size_t total = (size_t)n_tiles * B * B;
    long long best = 0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t rc = (uint32_t)(idx % (B * B));
        const uint32_t t = tile_ids ? tile_ids[idx / (B * B)] : (uint32_t)(idx / (B * B));
        long long v = acc[(size_t)t * B * B + rc];
        if (tile_row[t] == tile_col[t]) {  // a diagonal tile holds each pair in either orientation
            const uint32_t r = rc / B, c = rc % B;
            v = r < c ? v + acc[(size_t)t * B * B + c * B + r] : 0;
        }
        best = max(best, v);
    }
    for (int off = 32; off > 0; off >>= 1) best = max(best, (long long)__shfl_down(best, off));
    __shared__ long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = max(max(part[0], part[1]), max(part[2], part[3]));
        // (double)int64 * 2^-k is monotone, so the max of the integers gives the max of the doubles
        atomicMax(out_bits, (unsigned long long)__double_as_longlong((double)best * scale));
    }
}

// mode: 0 ADD_MIN, 1 EXPONENTIATE, 2 SCALE_MAX_1, 3 raw D (reference: similarity_matrix.cpp:271-293).
// One workgroup per 32 x 32 sub-block of an (upper-triangular) accumulator tile: the sub-block is
// read once with coalesced loads, normalised into LDS, and written twice -- as out[i][j] straight,
// as out[j][i] transposed through LDS -- so both the reads and the writes are full 256-byte rows.
template <int B>
// Only rows [row_begin, row_end) are written, to out[(i - row_begin) * n + j]: a rank that keeps a
// row block of the matrix (BASELINE config 5) passes its range, everyone else [0, n).
__global__ __launch_bounds__(256) void write_matrix(const long long *acc, const uint16_t *tile_row,
                                                    const uint16_t *tile_col, uint32_t n, double scale, int mode,
                                                    const unsigned long long *max_bits, uint32_t row_begin,
                                                    uint32_t row_end, double *out) {
    constexpr uint32_t SB = 32, PER = B / SB;
    __shared__ double V[SB][SB + 1];
    const uint32_t t = blockIdx.x / (PER * PER), sub = blockIdx.x % (PER * PER);
    const uint32_t a = sub / PER, b = sub % PER;
    const uint32_t I = tile_row[t], J = tile_col[t];
    const bool diag = (I == J);
    if (diag && a > b) return;  // covered by the mirror of (b, a)
    const double mx = (mode == 0 || mode == 2) ? __longlong_as_double((long long)*max_bits) : 0.0;
    // ADD_MIN: sim = -D; sim += |min(sim)|, and min(sim) = -max(D) with the zero diagonal included
    // SCALE_MAX_1: sim = D * (1 / max(D)) with the zero diagonal included (1/0 = inf as in the reference)
    const double add = fabs(-mx);
    const double inv = 1.0 / mx;
    const long long *tile = acc + (size_t)t * B * B;
    const uint32_t tx = threadIdx.x % SB, ty = threadIdx.x / SB;  // 32 x 8
    const uint32_t i0 = I * B + a * SB, j0 = J * B + b * SB;
    // rows i0.. (straight) and j0.. (mirror) both outside the range: nothing to do
    if ((i0 >= row_end || i0 + SB <= row_begin) && (j0 >= row_end || j0 + SB <= row_begin)) return;
#pragma unroll
    for (uint32_t k = 0; k < SB; k += 8) {
        const uint32_t r = ty + k, c = tx;
        long long v = tile[(a * SB + r) * B + b * SB + c];
        if (diag) v += tile[(b * SB + c) * B + a * SB + r];
        const double d = (double)v * scale;
        double w;
        switch (mode) {
            case 0: w = (d * -1.0) + add; break;
            case 1: w = 1.0 / (exp(d) + 1.0); break;
            case 2: w = d * inv; break;
            default: w = d; break;
        }
        if (i0 + r == j0 + c) w = 0.0;
        V[r][c] = w;
        const uint32_t i = i0 + r;
        if (i < n && j0 + c < n && i >= row_begin && i < row_end) out[(size_t)(i - row_begin) * n + j0 + c] = w;
    }
    if (diag && a == b) return;  // the sub-block is symmetric in itself
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < SB; k += 8) {
        const uint32_t c = ty + k, r = tx;  // out[j0 + c][i0 + r], r fastest
        const uint32_t j = j0 + c;
        if (i0 + r < n && j < n && j >= row_begin && j < row_end) out[(size_t)(j - row_begin) * n + i0 + r] = V[r][c];
    }
}

template <int B, int THREADS, int CAPJ, int CAPL, int HCAP, bool MASKS, bool COUNTS>
hipError_t launch_acc(const AccumulateArgs &args, uint32_t grid, hipStream_t stream) {
    constexpr size_t lds = (size_t)B * B * (COUNTS ? 4 : 8) + (size_t)CAPJ * 2 + ((size_t)CAPL + 2) * 2
            + (MASKS ? (size_t)CAPJ * 4 + SLUT_DIM * SLUT_DIM * 8 : 0) + 16
            + (size_t)(THREADS / 64) * ((size_t)HCAP + 64 * 8 + (MASKS ? 64 * 4 : 0)
                                        + (size_t)joint_list_cap<B, THREADS, CAPJ, CAPL, HCAP, MASKS, COUNTS>() * 10);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((CAPL + 2) % 4 == 0 && CAPJ % 8 == 0 && HCAP % 16 == 0, "alignment of the LDS carve-up");
    auto kern = &accumulate_tiles<B, THREADS, CAPJ, CAPL, HCAP, MASKS, COUNTS>;
    static thread_local int configured_device = -1;  // the attribute is per device and sticky
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (configured_device != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return e;
        configured_device = dev;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, stream, args);
    hipLaunchKernelGGL((reduce_slabs<B, COUNTS>), dim3(args.n_tiles * (B * B / 256)), dim3(256), 0, stream,
                       args.slab, args.tile_wg_begin, args.tile_begin, args.tile_ids, args.lut,
                       reinterpret_cast<long long *>(args.acc));
    return hipGetLastError();
}

template <int B>
hipError_t launch_correct(const AccumulateArgs &args, hipStream_t stream, const SideStream *side);

template <int B, int THREADS, int CAPJ, int CAPL, int GROUP>
hipError_t launch_masks(const AccumulateArgs &args, uint32_t grid, hipStream_t stream) {
    constexpr size_t lds = (size_t)B * MASKS_ROW_Q(B) * 8 + (size_t)CAPJ * 8 + ((size_t)CAPL + 2) * 2
            + MLUT_WORDS * 8 + (size_t)(THREADS / 64) * MASKS_RING * 12;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static_assert((CAPL + 2) % 4 == 0 && CAPJ % 8 == 0, "alignment of the LDS carve-up");
    auto kern = &accumulate_masks<B, THREADS, CAPJ, CAPL, GROUP>;
    static thread_local int configured_device = -1;  // the attribute is per device and sticky
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (configured_device != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return e;
        configured_device = dev;
    }
    static const bool slot_asm = [] { const char *v = std::getenv("SECEDO_MASKS_SLOT_ASM"); return !(v && std::atoi(v) == 0); }();
    AccumulateArgs with_flag = args;
    with_flag.masks_slot_asm = slot_asm ? 1u : 0u;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, stream, with_flag);
    hipLaunchKernelGGL((reduce_slabs<B, false>), dim3(args.n_tiles * (B * B / 256)), dim3(256), 0, stream, args.slab,
                       args.tile_wg_begin, args.tile_begin, args.tile_ids, args.lut,
                       reinterpret_cast<long long *>(args.acc));
    if (args.wide_list) {  // the pairs of reads that reach beyond their windows
        WideArgs w;
        w.blk_off = args.blk_off;
        w.stride = args.stride;
        w.entry32 = args.entry32;
        w.entry = args.entry;
        w.slow = args.slow;
        w.lut = args.lut;
        w.tile_row = args.tile_row;
        w.tile_col = args.tile_col;
        w.tile_begin = args.tile_begin;
        w.tile_ids = args.tile_ids;
        w.wide_off = args.wide_off;
        w.wide_list = args.wide_list;
        w.acc = reinterpret_cast<long long *>(args.acc);
        w.counters = args.counters;
        hipLaunchKernelGGL((wide_pairs<B>), dim3(args.n_tiles), dim3(256), 0, stream, w);
    }
    return hipGetLastError();
}

template <int B, int THREADS, int CAPJ, int CAPL, int GROUP, bool SLOT_ASM>
hipError_t launch_counts_v(const AccumulateArgs &args, uint32_t grid, hipStream_t stream, const SideStream *side,
                           hipEvent_t mid) {
    constexpr size_t lds = ((size_t)B * (B + 1) * 4 + 15) / 16 * 16 + (size_t)CAPJ * 4 + ((size_t)CAPL + 2) * 2
            + (size_t)(THREADS / 64) * (size_t)COUNTS_RING * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((CAPL + 2) % 4 == 0 && CAPJ % 8 == 0, "alignment of the LDS carve-up");
    auto kern = &accumulate_counts<B, THREADS, CAPJ, CAPL, GROUP, SLOT_ASM>;
    static thread_local int configured_device = -1;  // the attribute is per device and sticky
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (configured_device != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return e;
        configured_device = dev;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, stream, args);
    if (mid && (e = hipEventRecord(mid, stream)) != hipSuccess) return e;
    if (side && side->deferred && (e = side->deferred(side->deferred_ctx)) != hipSuccess) return e;
    return launch_correct<B>(args, stream, side);
}

// (SECEDO_SLOT_ASM=0: the compiler's pair slot, for A/B measurements)
template <int B, int THREADS, int CAPJ, int CAPL, int GROUP>
hipError_t launch_counts(const AccumulateArgs &args, uint32_t grid, hipStream_t stream, const SideStream *side,
                         hipEvent_t mid) {
    static const bool slot_asm = [] { const char *e = std::getenv("SECEDO_SLOT_ASM"); return !(e && std::atoi(e) == 0); }();
    return slot_asm ? launch_counts_v<B, THREADS, CAPJ, CAPL, GROUP, true>(args, grid, stream, side, mid)
                    : launch_counts_v<B, THREADS, CAPJ, CAPL, GROUP, false>(args, grid, stream, side, mid);
}

// The second kernel of the sparse-loci path, after the pair kernel.
template <int B>
hipError_t launch_correct(const AccumulateArgs &args, hipStream_t stream, const SideStream *side) {
    hipError_t e = hipSuccess;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    // What the flags of the reads mean, and the count tiles' way into the accumulator: one workgroup per tile.
    // The flagged entries' lists may still be in the making on the side stream (build_flagged_lists).
    if (side && side->stream) {
        if ((e = hipStreamWaitEvent(stream, side->join, 0)) != hipSuccess) return e;
    }
    CorrectArgs c;
    c.blk_off = args.blk_off;
    c.stride = args.stride;
    c.flag_grp = args.flag_grp;
    c.flag_rec = args.flag_rec;
    c.flag_idx = args.flag_idx;
    c.slow = args.slow;
    c.lut = args.lut;
    c.tile_row = args.tile_row;
    c.tile_col = args.tile_col;
    c.tile_begin = args.tile_begin;
    c.tile_ids = args.tile_ids;
    c.slab = args.slab;
    c.tile_wg_begin = args.tile_wg_begin;
    c.acc = args.acc;
    c.counters = args.counters;
    constexpr int CT = B == 128 ? 1024 : 256;
    constexpr size_t corr_lds = (size_t)B * B * 8;
    auto corr = &correct_tiles<B, CT>;
    static thread_local int corr_device = -1;
    if (corr_device != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(corr), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)corr_lds);
        if (e != hipSuccess) return e;
        corr_device = dev;
    }
    // (few tiles: several workgroups each; they add with atomics, into zeroes when the launch is to overwrite
    // -- a contiguous tile range then)
    c.split = counts_split(args.n_tiles);
    if (args.overwrite && args.tile_ids) c.split = 1;
    c.overwrite = (args.overwrite && c.split == 1u) ? 1u : 0u;
    c.max_bits = c.split == 1u ? args.max_bits : nullptr;
    c.max_scale = args.max_scale;
    if (args.overwrite && c.split > 1u) {
        e = hipMemsetAsync(args.acc + (size_t)args.tile_begin * B * B, 0, (size_t)args.n_tiles * B * B * 8, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(corr, dim3(args.n_tiles * c.split), dim3(CT), corr_lds, stream, c);
    return hipGetLastError();
}

// acc[tile] = 0 for the listed tiles (16 bytes per thread)
__global__ __launch_bounds__(256) void zero_tiles(long long *acc, const uint32_t *tile_ids, uint32_t n_tiles,
                                                 uint32_t tile_elems) {
    for (uint32_t t = blockIdx.y; t < n_tiles; t += gridDim.y) {
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(acc + (size_t)tile_ids[t] * tile_elems);
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < tile_elems / 2; i += gridDim.x * 256)
            dst[i] = make_ulonglong2(0ull, 0ull);
    }
}

int pair_mode() {
    static const int mode = [] {
        const char *e = std::getenv("SECEDO_PAIR_MODE");
        return e ? std::atoi(e) : 1;
    }();
    return mode;
}

}  // namespace

bool counts_path_enabled() { return pair_mode() != 0; }

// (few tiles: several workgroups each, as many as fit one round of the chip's 256 CUs)
uint32_t counts_split(uint32_t n_tiles) { return std::max(1u, std::min(8u, 256u / std::max(n_tiles, 1u))); }

size_t flagged_scan_bytes(uint32_t n_entries) {
    size_t bytes = 0;
    hipcub::CountingInputIterator<uint32_t> ids(0u);
    hipcub::TransformInputIterator<uint32_t, FlaggedOp, hipcub::CountingInputIterator<uint32_t>> in(
            ids, FlaggedOp{nullptr, 0});
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, static_cast<uint32_t *>(nullptr), (int)n_entries + 1);
    return bytes;
}

hipError_t build_flagged_lists(const uint32_t *entry32, const uint4 *entry, uint32_t n_entries,
                               const uint32_t *blk_off, size_t n_off, void *scan_tmp, size_t scan_tmp_bytes,
                               uint32_t *pre, uint32_t *grp, uint4 *rec, uint32_t *idx, hipStream_t stream) {
    hipcub::CountingInputIterator<uint32_t> ids(0u);
    hipcub::TransformInputIterator<uint32_t, FlaggedOp, hipcub::CountingInputIterator<uint32_t>> in(
            ids, FlaggedOp{entry32, n_entries});
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_tmp_bytes, in, pre, (int)n_entries + 1, stream);
    if (e != hipSuccess) return e;
    if (n_entries) {
        const uint32_t blocks = (uint32_t)std::min<size_t>(((size_t)n_entries + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(flagged_compact, dim3(blocks), dim3(256), 0, stream, entry32, entry, n_entries, pre, rec, idx);
    }
    if (n_off) {
        const uint32_t blocks = (uint32_t)std::min<size_t>((n_off + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(flagged_groups, dim3(blocks), dim3(256), 0, stream, blk_off, n_off, pre, grp);
    }
    return hipGetLastError();
}

hipError_t masks_words(const uint32_t *entry32, const uint32_t *mask32, uint32_t n_entries, uint32_t *y, uint32_t *xcol,
                       uint32_t *xrow, hipStream_t stream) {
    if (n_entries == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<size_t>(((size_t)n_entries + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_masks_words, dim3(blocks), dim3(256), 0, stream, entry32, mask32, n_entries, y, xcol, xrow);
    return hipGetLastError();
}

hipError_t wide_count(const uint32_t *entry32, const uint32_t *blk_off, uint32_t stride, uint32_t num_blocks,
                      uint32_t *cnt, uint32_t *off, uint32_t *cur, hipStream_t stream) {
    if (num_blocks == 0 || num_blocks > 1024) return hipErrorInvalidValue;  // (the caller keeps such pileups off this path)
    hipError_t e = hipMemsetAsync(cnt, 0, (size_t)num_blocks * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_wide_count, dim3(16, num_blocks), dim3(256), 0, stream, entry32, blk_off, stride, cnt);
    hipLaunchKernelGGL(k_wide_scan, dim3(1), dim3(1024), 0, stream, cnt, num_blocks, off, cur);
    return hipGetLastError();
}

hipError_t wide_fill(const uint32_t *entry32, const uint32_t *blk_off, uint32_t stride, uint32_t num_blocks,
                     uint32_t *cur, uint32_t *list, hipStream_t stream) {
    hipLaunchKernelGGL(k_wide_fill, dim3(16, num_blocks), dim3(256), 0, stream, entry32, blk_off, stride, cur, list);
    return hipGetLastError();
}

size_t accumulate_slab_bytes(uint32_t block_cells, bool count_tile, uint32_t n_workgroups) {
    return (size_t)n_workgroups * block_cells * block_cells * (count_tile ? 4 : 8);
}

StageGeometry stage_geometry(uint32_t block_cells) {
    if (block_cells == 128)
        return StageGeometry{kCapJ128, kCapL128, kCapJ128, kCapL128, kCapJ128C, kCapL128C, 2.0 /* never */};
    return StageGeometry{kCapJ64, kCapL64, kCapJ64M, kCapL64M, kCapJ64C, kCapL64C, kMasksThreshold};
}

hipError_t launch_accumulate(const AccumulateArgs &args, uint32_t block_cells, bool stage_masks,
                             bool count_tile, uint32_t n_tiles, hipStream_t stream, const SideStream *side,
                             hipEvent_t mid) {
    if (n_tiles == 0) return hipSuccess;
    const uint32_t grid = args.n_workgroups;
    const bool counts_path = count_tile && !stage_masks && pair_mode() != 0;
    if (args.overwrite && !counts_path) {  // (accumulate_counts + correct_tiles store the tiles themselves)
        const uint32_t tile_elems = block_cells * block_cells;
        if (args.tile_ids) {
            hipLaunchKernelGGL(zero_tiles, dim3(8, std::min(n_tiles, 65535u)), dim3(256), 0, stream,
                               reinterpret_cast<long long *>(args.acc), args.tile_ids, n_tiles, tile_elems);
        } else {
            const hipError_t e = hipMemsetAsync(args.acc + (size_t)args.tile_begin * tile_elems, 0,
                                                (size_t)n_tiles * tile_elems * 8, stream);
            if (e != hipSuccess) return e;
        }
    }
    if (block_cells == 128) {
        // the 128 KiB int64 tile leaves no room for the window masks: joint terms go through HBM
        if (count_tile && pair_mode() != 0) {
            static const int g = [] { const char *e = std::getenv("SECEDO_GROUP"); return e ? std::atoi(e) : 0; }();
            if (g == 1) return launch_counts<128, 1024, kCapJ128C, kCapL128C, 1>(args, grid, stream, side, mid);
            if (g == 5) return launch_counts<128, 1024, kCapJ128C, kCapL128C, 5>(args, grid, stream, side, mid);
            if (g == 6) return launch_counts<128, 1024, kCapJ128C, kCapL128C, 6>(args, grid, stream, side, mid);
            if (g == 8) return launch_counts<128, 1024, kCapJ128C, kCapL128C, 8>(args, grid, stream, side, mid);
            if (g == 3) return launch_counts<128, 1024, kCapJ128C, kCapL128C, 3>(args, grid, stream, side, mid);
            if (g == 4) return launch_counts<128, 1024, kCapJ128C, kCapL128C, 4>(args, grid, stream, side, mid);
            if (g == 2 || (g == 0 && args.group_hint == 2))
                return launch_counts<128, 1024, kCapJ128C, kCapL128C, 2>(args, grid, stream, side, mid);
            if (g == 0 && args.group_hint == 3)
                return launch_counts<128, 1024, kCapJ128C, kCapL128C, 3>(args, grid, stream, side, mid);
            return launch_counts<128, 1024, kCapJ128C, kCapL128C, 4>(args, grid, stream, side, mid);
        }
        if (count_tile) return launch_acc<128, 1024, kCapJ128C, kCapL128C, 1024, false, true>(args, grid, stream);
        return launch_acc<128, 1024, kCapJ128, kCapL128, 512, false, false>(args, grid, stream);
    }
    if (stage_masks && args.masks_kernel) {
        // column entries per item and pass: C2 clustered (9 entries per cell block and locus) 1.160 / 1.143 / 1.111 ms
        // of accumulate with 4 / 2 / 3
        static const int g = [] { const char *e = std::getenv("SECEDO_MASKS_GROUP"); return e ? std::atoi(e) : 3; }();
        if (g == 2) return launch_masks<64, 512, kCapJ64M, kCapL64M, 2>(args, grid, stream);
        if (g == 4) return launch_masks<64, 512, kCapJ64M, kCapL64M, 4>(args, grid, stream);
        return launch_masks<64, 512, kCapJ64M, kCapL64M, 3>(args, grid, stream);
    }
    if (stage_masks) return launch_acc<64, 512, kCapJ64M, kCapL64M, 1024, true, false>(args, grid, stream);
    if (count_tile && pair_mode() != 0) return launch_counts<64, 512, kCapJ64C, kCapL64C, 4>(args, grid, stream, side, mid);
    if (count_tile) return launch_acc<64, 256, kCapJ64C, kCapL64C, 1024, false, true>(args, grid, stream);
    return launch_acc<64, 256, kCapJ64, kCapL64, 1024, false, false>(args, grid, stream);
}

__global__ void k_add_terms(long long *acc, const unsigned long long *index, const long long *value, uint32_t n) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + index[k], (unsigned long long)value[k]);
}

hipError_t launch_add_terms(int64_t *acc, const unsigned long long *index, const long long *value, uint32_t n,
                            hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_terms, dim3((n + 255) / 256), dim3(256), 0, stream, reinterpret_cast<long long *>(acc), index,
                       value, n);
    return hipGetLastError();
}

hipError_t launch_tile_max(const int64_t *acc, const uint16_t *tile_row, const uint16_t *tile_col,
                           const uint32_t *tile_ids, uint32_t n_tiles, uint32_t block_cells, int scale_log2,
                           unsigned long long *d_max_bits, hipStream_t stream) {
    const double scale = ldexp(1.0, -scale_log2);
    hipError_t e = hipMemsetAsync(d_max_bits, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess || n_tiles == 0) return e;
    const size_t total = (size_t)n_tiles * block_cells * block_cells;
    // every workgroup ends in one atomicMax on the same word, and same-address atomics are served one
    // after the other (~10 ns each): at least 16 cells per thread
    const uint32_t grid = (uint32_t)std::max<size_t>(1, std::min<size_t>((total + 256 * 16 - 1) / (256 * 16), 256 * 8));
    const long long *a = reinterpret_cast<const long long *>(acc);
    if (block_cells == 128) {
        hipLaunchKernelGGL((reduce_max<128>), dim3(grid), dim3(256), 0, stream, a, tile_row, tile_col, tile_ids, n_tiles,
                           scale, d_max_bits);
    } else {
        hipLaunchKernelGGL((reduce_max<64>), dim3(grid), dim3(256), 0, stream, a, tile_row, tile_col, tile_ids, n_tiles,
                           scale, d_max_bits);
    }
    return hipGetLastError();
}

// keep_max: d_max_bits already holds the maximum to normalise with (e.g. reduced over ranks)
hipError_t launch_finalize(const int64_t *acc, const uint16_t *tile_row, const uint16_t *tile_col, uint32_t n_tiles,
                           uint32_t n, uint32_t block_cells, int scale_log2, int mode,
                           unsigned long long *d_max_bits, uint32_t row_begin, uint32_t row_end, double *out,
                           hipStream_t stream, bool keep_max) {
    const double scale = ldexp(1.0, -scale_log2);
    const long long *a = reinterpret_cast<const long long *>(acc);
    if (n_tiles == 0) return hipSuccess;
    if ((mode == 0 || mode == 2) && !keep_max) {
        hipError_t e = launch_tile_max(acc, tile_row, tile_col, nullptr, n_tiles, block_cells, scale_log2, d_max_bits,
                                       stream);
        if (e != hipSuccess) return e;
    }
    if (block_cells == 128) {
        hipLaunchKernelGGL((write_matrix<128>), dim3(n_tiles * 16), dim3(256), 0, stream, a, tile_row, tile_col, n,
                           scale, mode, d_max_bits, row_begin, row_end, out);
    } else {
        hipLaunchKernelGGL((write_matrix<64>), dim3(n_tiles * 4), dim3(256), 0, stream, a, tile_row, tile_col, n,
                           scale, mode, d_max_bits, row_begin, row_end, out);
    }
    return hipGetLastError();
}

}  // namespace secedo
