// simmat_api.cpp -- implementation of the C-ABI declared in include/secedo_simmat.h.
//
// Host orchestration only: argument checks, host packing (pack_host.cpp), HBM buffers, kernel
// launches (simmat_kernels.hip). There is deliberately no CPU compute path in here: if no HIP
// device is usable the entry points fail with SECEDO_E_NO_DEVICE.
#include "secedo_simmat.h"

#include "filter_device.hpp"
#include "filter_host.hpp"
#include "llr_table.hpp"
#include "pack_device.hpp"
#include "pack_host.hpp"
#include "simmat_kernels.hpp"

#include <hip/hip_runtime_api.h>
#if defined(__linux__)
#include <sys/mman.h>
#endif

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <queue>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return fail(SECEDO_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);  \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e == hipSuccess) bytes = n ? n : 16;
        if (e == hipSuccess && secedo::poison_level() >= 1) e = hipMemset(p, 0xA5, bytes);
        return e;
    }
    template <class T>
    hipError_t upload(const std::vector<T> &v) {
        hipError_t e = ensure(v.size() * sizeof(T));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

// Small host -> device uploads in the ORDER OF A STREAM (tables, workgroup plans, tile lists of accumulate): the
// bytes are copied into one of four host slots of the handle first, so that the caller's / the builder's memory
// is free at once and the copy may execute whenever the stream gets to it; a slot is taken again only after the
// copy that used it last has executed (its event), which blocks the host only when five uploads are in flight.
struct StagedUploads {
    static constexpr int kSlots = 4;
    std::vector<unsigned char> host[kSlots];
    hipEvent_t done[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool used[kSlots] = {false, false, false, false};
    int next = 0;
    hipError_t put(void *dst, const void *src, size_t bytes, hipStream_t s) {
        if (bytes == 0) return hipSuccess;
        const int k = next;
        next = (next + 1) % kSlots;
        hipError_t e = hipSuccess;
        if (done[k] && used[k] && hipEventSynchronize(done[k]) != hipSuccess) {
            // (the stream it was recorded on is gone -- destroyed streams finish their work first: a fresh event)
            (void)hipGetLastError();
            (void)hipEventDestroy(done[k]);
            done[k] = nullptr;
        }
        used[k] = false;
        if (!done[k]) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
        if (e != hipSuccess) return e;
        host[k].assign(static_cast<const unsigned char *>(src), static_cast<const unsigned char *>(src) + bytes);
        e = hipMemcpyAsync(dst, host[k].data(), bytes, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipEventRecord(done[k], s);
        used[k] = (e == hipSuccess);
        return e;
    }
    // every upload so far has executed: called before the stream they were ordered on is destroyed
    void settle() {
        for (int k = 0; k < kSlots; ++k) {
            if (done[k] && used[k] && hipEventSynchronize(done[k]) != hipSuccess) (void)hipGetLastError();
            used[k] = false;
        }
    }
    void destroy() {
        for (int k = 0; k < kSlots; ++k) {
            if (done[k]) {
                if (used[k]) (void)hipEventSynchronize(done[k]);
                (void)hipEventDestroy(done[k]);
            }
            done[k] = nullptr;
            used[k] = false;
        }
    }
};

}  // namespace

struct secedo_simmat {
    int device = 0;
    // the pileup handed to set_pileup / set_pileup_device (borrowed until prepare returns)
    secedo::FlatPileupView view;         // host pointers
    secedo::DeviceFlatPileup dview;      // device pointers
    bool have_host = false, have_device = false;
    bool prepared = false;
    int packing_mode = 0;                // 0 auto (device, host when required), 1 host, 2 device only
    uint32_t num_threads = 1;            // of the last prepare (the reference's parameter; bounds helper threads)
    int used_device_packing = 0;

    // raw pileup uploaded by prepare() when it came as host pointers
    DevBuf raw_chr, raw_pos, raw_off, raw_rid, raw_idb, raw_g2p;

    // the packed pileup in HBM + geometry
    secedo::DevicePacked pk;
    uint32_t num_tiles = 0;
    DevBuf tile_row, tile_col, lut, counters, max_bits, slow_args, slab, plan_wg_tile, plan_wg_begin;
    uint32_t plan_tile_begin = 0xFFFFFFFFu, plan_tile_end = 0, plan_ranges = 0, plan_blocks = 0, plan_workgroups = 0;
    DevBuf flag_tmp, flag_pre, flag_grp, flag_rec, flag_idx;  // sparse-loci path: the flagged entries, compact
    bool flags_ready = false;                                 // ... of the current packed pileup
    bool wide_known = false;                                  // clustered loci: the reads that reach beyond their windows ...
    uint32_t n_wide = 0;                                      // ... their entries, listed per cell block
    DevBuf wide_tab, wide_list;
    DevBuf mk_words;                                          // ... the entries' words for accumulate_masks (y | xcol | xrow)
    DevBuf own_acc, own_out;  // used by the one-shot entry point only
    DevBuf tile_ids;          // tile list of accumulate_list / max_of_tiles
    std::vector<uint16_t> host_tile_row, host_tile_col;
    StagedUploads uploads;    // tables, plans and tile lists of accumulate, in the order of its stream
    // read pairs that share more than 128 loci: noted by the kernels, evaluated as the reference does on the host
    DevBuf beyond_list, beyond_count, beyond_index, beyond_value;
    uint64_t plan_list_hash = 0;  // 0: the cached workgroup plan belongs to a contiguous tile range

    // LLR table of the last accumulate()
    bool have_model = false, have_lut = false, have_slow = false;
    double lut_eps = 0, lut_h = 0, lut_theta = 0;
    int scale_log2 = 44;
    int scale_wanted = 44;               // llr_scale_for(table, scale_for_bound, scale_for_reach)
    uint64_t scale_for_bound = ~0ull;
    uint32_t scale_for_reach = ~0u;
    // secedo_simmat_set_pair_bound / set_scale_bounds: the bounds of everything that is summed into one
    // accumulator (shards on several ranks); they belong to the pileup that was set when they were given
    uint64_t pair_bound_override = 0;
    uint32_t max_shared_override = 0;
    uint64_t pileup_identity = 0, override_identity = 0;
    bool override_dropped = false;  // bounds were in force and another pileup took them away (scale_bounds_state 2)
    secedo::LlrModel model;
    secedo::LlrTable table;
    secedo::SlowPathArgs slow_host;

    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_mid = nullptr;
    bool timed_mid = false;
    bool timed = false;
    secedo::SideStream side;  // the flagged entries' lists are built beside accumulate_counts (created on first use)
};

namespace {

template <class T>
hipError_t arena_upload(secedo::DeviceArena &a, const std::vector<T> &v) {
    hipError_t e = a.ensure(v.size() * sizeof(T));
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpy(a.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

template <class T>
hipError_t buf_upload(DevBuf &b, const T *src, size_t n) {
    hipError_t e = b.ensure(n * sizeof(T));
    if (e != hipSuccess || n == 0) return e;
    return hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace

namespace {

// Handles kept for secedo_simmat_compute, one per device, handed out to one caller at a time (a second
// concurrent caller on the same device gets a fresh handle). Deliberately not destroyed at process
// exit: static destructors would run after the HIP runtime is gone; the driver reclaims the memory.
// secedo_simmat_release_cache() frees them on request. SECEDO_ONE_SHOT_CACHE=0 turns the pool off.
std::mutex g_pool_mutex;
std::map<int, secedo_simmat_t *> &g_pool = *new std::map<int, secedo_simmat_t *>();

bool pool_enabled() {
    const char *env = std::getenv("SECEDO_ONE_SHOT_CACHE");
    return !(env && std::atoi(env) == 0);
}

// key: the device, or -- for the lanes of a multi-device call -- a number of its own per lane (kLaneKey)
constexpr int kLaneKey = 1 << 16;
int one_shot_acquire(int key, int device, secedo_simmat_t **h) {
    if (pool_enabled()) {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        auto it = g_pool.find(key);
        if (it != g_pool.end() && it->second) {
            *h = it->second;
            it->second = nullptr;
            return SECEDO_OK;
        }
    }
    return secedo_simmat_create(h, device);
}

void one_shot_release(int key, secedo_simmat_t *h, bool ok) {
    if (!h) return;
    if (ok && pool_enabled()) {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        secedo_simmat_t *&slot = g_pool[key];
        if (!slot) {
            slot = h;
            return;
        }
    }
    secedo_simmat_destroy(h);
}

}  // namespace

namespace secedo {
// for the other translation units of the library (spectral_api.cpp): one error slot per thread
int api_fail(int code, const std::string &msg) { return fail(code, msg); }
void em_release_cache();  // em_device.hip
}  // namespace secedo

// ---- pinned host memory kept between one-shot calls: the caller's staging (five buffers) and the bounce ring of
// the matrix download
namespace {
struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        const size_t want = n + n / 8 + 4096;  // some slack: the next sub-cluster is rarely the same size
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};
std::mutex g_staging_mutex;
bool g_staging_busy = false;
PinnedBuf g_staging[5];
// page-locked rings of the matrix download: [0] the single-device call's, [1 + k] lane k's of a multi-device call;
// each guarded by its mutex for the duration of a download
constexpr int kMaxLanes = 16;
struct Bounce {
    PinnedBuf buf;
    std::mutex mutex;
};
Bounce g_bounce[1 + kMaxLanes];

// d_src (device) -> dst (pageable host memory, typically the fresh pages of the caller's matrix): chunks go
// through a page-locked ring by DMA while `threads` host threads copy the chunk before out of the ring, each
// its own slice -- touching the destination's fresh pages in parallel is what a single-threaded copy into
// pageable memory cannot do (12-20 GB/s for hipMemcpy on the 512 MB of C3).
hipError_t download_pipelined(const void *d_src, void *dst, size_t bytes, unsigned threads, int ring = 0) {
    constexpr size_t kChunk = 32u << 20;
    constexpr int kSlots = 4;
    std::lock_guard<std::mutex> lock(g_bounce[ring].mutex);
    PinnedBuf &g_bounce = ::g_bounce[ring].buf;
    hipError_t e = g_bounce.ensure(kChunk * kSlots);
    if (e != hipSuccess) {  // no pinned memory to be had: the plain copy
        (void)hipGetLastError();
        return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost);
    }
#if defined(__linux__)
    {   // huge pages for the destination where the kernel grants them on request: 2 MiB faults instead of 4 KiB
        const uintptr_t a = (reinterpret_cast<uintptr_t>(dst) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const uintptr_t b = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (b > a) (void)madvise(reinterpret_cast<void *>(a), b - a, MADV_HUGEPAGE);
    }
#endif
    // (SECEDO_DOWNLOAD_STREAMS=2: chunks alternate between two streams -- measured SLOWER on the MI355X box, 17-23 ms
    // against 14-15 for the 512 MB of C3: one stream moves 34-36 GB/s and a second one only gets in its way)
    static const int n_streams = [] { const char *v = std::getenv("SECEDO_DOWNLOAD_STREAMS"); const int n = v ? std::atoi(v) : 1; return n < 1 ? 1 : n > 2 ? 2 : n; }();
    hipStream_t s = nullptr, s2 = nullptr;
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) return e;
    if (n_streams > 1 && (e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)) != hipSuccess) {
        (void)hipStreamDestroy(s);
        return e;
    }
    for (int i = 0; i < kSlots && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    threads = std::max(1u, std::min(threads, 16u));
    auto issue = [&](size_t c) {
        const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
        hipStream_t sc = (s2 && (c & 1)) ? s2 : s;
        hipError_t r = hipMemcpyAsync(static_cast<char *>(g_bounce.p) + (c % kSlots) * kChunk,
                                      static_cast<const char *>(d_src) + off, len, hipMemcpyDeviceToHost, sc);
        if (r == hipSuccess) r = hipEventRecord(ev[c % kSlots], sc);
        return r;
    };
    // The copying threads live for the whole download (round 4: a team per chunk was 16 x 7 thread starts and joins,
    // a fifth of the 14 ms of a C3 matrix): thread t copies slice t of every chunk as soon as the chunk has landed
    // (`landed`, published by this thread after the chunk's event), and a ring slot is written again only when every
    // thread is done with the chunk that used it (`copied`).
    std::atomic<size_t> landed{0};
    std::atomic<bool> give_up{false};
    std::vector<std::atomic<unsigned>> copied(n_chunks);
    for (auto &c : copied) c.store(0);
    auto copier = [&](unsigned t) {
        for (size_t c = 0; c < n_chunks; ++c) {
            while (landed.load(std::memory_order_acquire) <= c) {
                if (give_up.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
            const char *src = static_cast<const char *>(g_bounce.p) + (c % kSlots) * kChunk;
            char *out = static_cast<char *>(dst) + off;
            const size_t slice = ((len + threads - 1) / threads + 4095) & ~(size_t)4095;
            const size_t lo = std::min(len, (size_t)t * slice), hi = std::min(len, ((size_t)t + 1) * slice);
            if (hi > lo) std::memcpy(out + lo, src + lo, hi - lo);
            copied[c].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(copier, t);
    for (size_t c = 0; c < std::min<size_t>(kSlots - 1, n_chunks) && e == hipSuccess; ++c) e = issue(c);
    for (size_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
        if (c + kSlots - 1 < n_chunks) {  // its slot held chunk c - 1
            while (c > 0 && copied[c - 1].load(std::memory_order_acquire) < threads) std::this_thread::yield();
            e = issue(c + kSlots - 1);
        }
        if (e == hipSuccess) e = hipEventSynchronize(ev[c % kSlots]);
        if (e != hipSuccess) break;
        landed.store(c + 1, std::memory_order_release);
    }
    if (e != hipSuccess) give_up.store(true);
    for (auto &th : pool) th.join();
    (void)hipStreamSynchronize(s);
    if (s2) (void)hipStreamSynchronize(s2);
    for (int i = 0; i < kSlots; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    (void)hipStreamDestroy(s);
    if (s2) (void)hipStreamDestroy(s2);
    return e;
}
}  // namespace

// Which pileup the handle holds, as far as the scale overrides care: sizes and the entry array. Bounds set for
// the shards of one pileup must not leak into the accumulation of another (they lower the fixed-point scale);
// packing the same arrays again (another tile edge, the next step of a loop) keeps them.
static void note_pileup(secedo_simmat *h, uint32_t n_chr, uint64_t n_loci, uint64_t n_entries, const void *read_ids) {
    uint64_t id = 0xcbf29ce484222325ull;
    for (uint64_t v : {(uint64_t)n_chr, n_loci, n_entries, (uint64_t)reinterpret_cast<uintptr_t>(read_ids)})
        id = (id ^ v) * 0x100000001b3ull;
    h->pileup_identity = id | 1ull;
    if (h->override_identity != h->pileup_identity) {
        if (h->pair_bound_override || h->max_shared_override) h->override_dropped = true;
        h->pair_bound_override = 0;
        h->max_shared_override = 0;
        h->override_identity = 0;
    }
}

extern "C" {

const char *secedo_simmat_last_error(void) { return g_last_error.c_str(); }

const char *secedo_simmat_version(void) { return "secedo-simmat-mi355x 0.1 (gfx950)"; }

int secedo_simmat_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int secedo_simmat_normalization_from_string(const char *name) {
    if (name) {
        if (!std::strcmp(name, "ADD_MIN")) return SECEDO_NORM_ADD_MIN;
        if (!std::strcmp(name, "EXPONENTIATE")) return SECEDO_NORM_EXPONENTIATE;
        if (!std::strcmp(name, "SCALE_MAX_1")) return SECEDO_NORM_SCALE_MAX_1;
    }
    return fail(SECEDO_E_INVALID_NORMALIZATION,
                std::string("Invalid normalization: ") + (name ? name : "(null)"));
}

double secedo_simmat_llr(uint32_t x_s, uint32_t x_d, double eps, double h, double theta) {
    if (x_s + x_d >= 1 && !secedo::llr_exact_mode()) return secedo::reference_llr_any(eps, h, theta, x_s, x_d, 8);
    return secedo::llr(secedo::make_llr_model(eps, h, theta), x_s, x_d);
}

double secedo_simmat_llr_closed_form(uint32_t x_s, uint32_t x_d, double eps, double h, double theta) {
    return secedo::llr(secedo::make_llr_model(eps, h, theta), x_s, x_d);
}

int secedo_simmat_create(secedo_simmat_t **handle, int device_id) {
    if (!handle) return fail(SECEDO_E_INVALID_ARG, "handle is null");
    *handle = nullptr;
    const int n = secedo_simmat_device_count();
    if (n <= 0) return fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the similarity-matrix path has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(SECEDO_E_NO_DEVICE, "device id out of range");
    HIP_TRY(hipSetDevice(device_id));
    secedo_simmat *h = new (std::nothrow) secedo_simmat();
    if (!h) return fail(SECEDO_E_LIMIT, "out of host memory");
    h->device = device_id;
    hipError_t e = hipEventCreate(&h->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_end);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_mid);
    if (e != hipSuccess) {
        delete h;
        return hip_fail(e, "hipEventCreate");
    }
    *handle = h;
    return SECEDO_OK;
}

void secedo_simmat_destroy(secedo_simmat_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
    if (h->side.fork) (void)hipEventDestroy(h->side.fork);
    if (h->side.join) (void)hipEventDestroy(h->side.join);
    if (h->side.stream) (void)hipStreamDestroy(h->side.stream);
    h->uploads.destroy();
    delete h;
}

int secedo_simmat_set_pileup(secedo_simmat_t *h, const uint32_t *chr_locus_off, uint32_t n_chr,
                             const uint32_t *locus_pos, const uint64_t *locus_entry_off,
                             const uint32_t *read_ids, const uint16_t *id_base16,
                             const uint32_t *id_base32, const uint32_t *group_id_to_pos,
                             uint32_t n_groups) {
    if (!h) return fail(SECEDO_E_INVALID_ARG, "handle is null");
    if (!chr_locus_off || !locus_entry_off) return fail(SECEDO_E_INVALID_ARG, "null offset arrays");
    if ((id_base16 != nullptr) == (id_base32 != nullptr))
        return fail(SECEDO_E_INVALID_ARG, "exactly one of id_base16 / id_base32 must be given");
    if (!group_id_to_pos && n_groups) return fail(SECEDO_E_INVALID_ARG, "group_id_to_pos is null");
    h->view.chr_locus_off = chr_locus_off;
    h->view.n_chr = n_chr;
    h->view.locus_pos = locus_pos;
    h->view.locus_entry_off = locus_entry_off;
    h->view.read_ids = read_ids;
    h->view.id_base16 = id_base16;
    h->view.id_base32 = id_base32;
    h->view.group_id_to_pos = group_id_to_pos;
    h->view.n_groups = n_groups;
    h->have_host = true;
    h->have_device = false;
    h->prepared = false;
    note_pileup(h, n_chr, h->view.n_loci(), h->view.n_entries(), read_ids);
    return SECEDO_OK;
}

int secedo_simmat_set_pileup_device(secedo_simmat_t *h, const uint32_t *d_chr_locus_off, uint32_t n_chr,
                                    const uint32_t *d_locus_pos, const uint64_t *d_locus_entry_off,
                                    const uint32_t *d_read_ids, const uint16_t *d_id_base16,
                                    const uint32_t *d_id_base32, const uint32_t *d_group_id_to_pos,
                                    uint32_t n_groups, uint32_t n_loci, uint64_t n_entries) {
    if (!h) return fail(SECEDO_E_INVALID_ARG, "handle is null");
    if (!d_chr_locus_off || !d_locus_entry_off) return fail(SECEDO_E_INVALID_ARG, "null offset arrays");
    if ((d_id_base16 != nullptr) == (d_id_base32 != nullptr))
        return fail(SECEDO_E_INVALID_ARG, "exactly one of id_base16 / id_base32 must be given");
    if (!d_group_id_to_pos && n_groups) return fail(SECEDO_E_INVALID_ARG, "group_id_to_pos is null");
    h->dview.chr_locus_off = d_chr_locus_off;
    h->dview.n_chr = n_chr;
    h->dview.locus_pos = d_locus_pos;
    h->dview.locus_entry_off = d_locus_entry_off;
    h->dview.read_ids = d_read_ids;
    h->dview.id_base16 = d_id_base16;
    h->dview.id_base32 = d_id_base32;
    h->dview.group_id_to_pos = d_group_id_to_pos;
    h->dview.n_groups = n_groups;
    h->dview.n_loci = n_loci;
    h->dview.n_entries = n_entries;
    h->have_device = true;
    h->have_host = false;
    h->prepared = false;
    note_pileup(h, n_chr, n_loci, n_entries, d_read_ids);
    return SECEDO_OK;
}

int secedo_simmat_set_packing(secedo_simmat_t *h, int mode) {
    if (!h) return fail(SECEDO_E_INVALID_ARG, "handle is null");
    if (mode < 0 || mode > 2) return fail(SECEDO_E_INVALID_ARG, "packing mode must be 0 (auto), 1 (host) or 2 (device)");
    h->packing_mode = mode;
    return SECEDO_OK;
}

int secedo_simmat_used_device_packing(const secedo_simmat_t *h) { return h ? h->used_device_packing : 0; }

int secedo_simmat_prepare(secedo_simmat_t *h, uint32_t num_cells, uint32_t max_fragment_length,
                          uint32_t num_threads, uint32_t block_cells, void *stream) {
    if (!h) return fail(SECEDO_E_INVALID_ARG, "handle is null");
    if (!h->have_host && !h->have_device) return fail(SECEDO_E_STATE, "set_pileup was not called");
    if (block_cells == 0) {
        if (const char *env = std::getenv("SECEDO_BLOCK_CELLS")) {
            const int v = std::atoi(env);
            if (v == 64 || v == 128) block_cells = static_cast<uint32_t>(v);
        }
    }
    bool allow_count_tile = true;  // SECEDO_COUNT_TILE=0 forces the int64 tile (diagnostics)
    if (const char *env = std::getenv("SECEDO_COUNT_TILE")) allow_count_tile = std::atoi(env) != 0;
    int mode = h->packing_mode;
    if (const char *env = std::getenv("SECEDO_PACKING")) {
        if (!std::strcmp(env, "host")) mode = 1;
        if (!std::strcmp(env, "device")) mode = 2;
    }
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    h->prepared = false;
    h->used_device_packing = 0;
    h->num_threads = num_threads;
    secedo::DevicePacked &pk = h->pk;

    bool need_host = (mode == 1);
    if (!need_host) {
        if (h->have_host) {  // raw pileup to HBM
            const secedo::FlatPileupView &v = h->view;
            const uint32_t L = v.n_loci();
            const uint64_t E = v.n_entries();
            HIP_TRY(buf_upload(h->raw_chr, v.chr_locus_off, (size_t)v.n_chr + 1));
            HIP_TRY(buf_upload(h->raw_pos, v.locus_pos, L));
            HIP_TRY(buf_upload(h->raw_off, v.locus_entry_off, (size_t)L + 1));
            HIP_TRY(buf_upload(h->raw_rid, v.read_ids, E));
            if (v.id_base16) HIP_TRY(buf_upload(h->raw_idb, v.id_base16, E));
            else HIP_TRY(buf_upload(h->raw_idb, v.id_base32, E));
            HIP_TRY(buf_upload(h->raw_g2p, v.group_id_to_pos, v.n_groups));
            h->dview.chr_locus_off = h->raw_chr.as<uint32_t>();
            h->dview.n_chr = v.n_chr;
            h->dview.locus_pos = h->raw_pos.as<uint32_t>();
            h->dview.locus_entry_off = h->raw_off.as<uint64_t>();
            h->dview.read_ids = h->raw_rid.as<uint32_t>();
            h->dview.id_base16 = v.id_base16 ? h->raw_idb.as<uint16_t>() : nullptr;
            h->dview.id_base32 = v.id_base16 ? nullptr : h->raw_idb.as<uint32_t>();
            h->dview.group_id_to_pos = h->raw_g2p.as<uint32_t>();
            h->dview.n_groups = v.n_groups;
            h->dview.n_loci = L;
            h->dview.n_entries = E;
        }
        const std::string err = secedo::pack_pileup_device(h->dview, num_cells, max_fragment_length,
                                                           num_threads, block_cells, &secedo::stage_geometry,
                                                           allow_count_tile, s, &pk, &need_host);
        if (!err.empty()) {
            h->have_host = h->have_device = false;
            return fail(err.find("hip") == 0 ? SECEDO_E_HIP : SECEDO_E_INVALID_ARG, err);
        }
        if (need_host && mode == 2) {
            h->have_host = h->have_device = false;
            return fail(SECEDO_E_LIMIT, "this pileup needs the host packing path (a read is longer than "
                                        "max_fragment_length, or a size limit of the device path)");
        }
        if (!need_host) h->used_device_packing = 1;
    }

    if (need_host) {
        // the exact sequential emulation on the host (reads split at flushes, any size)
        std::vector<uint32_t> hc, hp, hr, hg, hi32;
        std::vector<uint64_t> ho;
        std::vector<uint16_t> hi16;
        secedo::FlatPileupView v = h->view;
        if (!h->have_host) {  // the pileup lives in HBM only: bring it back
            const secedo::DeviceFlatPileup &d = h->dview;
            HIP_TRY(hipStreamSynchronize(s));
            hc.resize((size_t)d.n_chr + 1);
            hp.resize(d.n_loci);
            ho.resize((size_t)d.n_loci + 1);
            hr.resize(d.n_entries);
            hg.resize(d.n_groups);
            HIP_TRY(hipMemcpy(hc.data(), d.chr_locus_off, hc.size() * 4, hipMemcpyDeviceToHost));
            if (d.n_loci) HIP_TRY(hipMemcpy(hp.data(), d.locus_pos, hp.size() * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(ho.data(), d.locus_entry_off, ho.size() * 8, hipMemcpyDeviceToHost));
            if (d.n_entries) HIP_TRY(hipMemcpy(hr.data(), d.read_ids, hr.size() * 4, hipMemcpyDeviceToHost));
            if (d.n_groups) HIP_TRY(hipMemcpy(hg.data(), d.group_id_to_pos, hg.size() * 4, hipMemcpyDeviceToHost));
            if (d.id_base16) {
                hi16.resize(d.n_entries);
                if (d.n_entries) HIP_TRY(hipMemcpy(hi16.data(), d.id_base16, hi16.size() * 2, hipMemcpyDeviceToHost));
            } else {
                hi32.resize(d.n_entries);
                if (d.n_entries) HIP_TRY(hipMemcpy(hi32.data(), d.id_base32, hi32.size() * 4, hipMemcpyDeviceToHost));
            }
            v.chr_locus_off = hc.data();
            v.n_chr = d.n_chr;
            v.locus_pos = hp.data();
            v.locus_entry_off = ho.data();
            v.read_ids = hr.data();
            // (an empty vector's data() may be null, and "which of the two" is told by the pointers)
            static const uint16_t none16 = 0;
            static const uint32_t none32 = 0;
            v.id_base16 = d.id_base16 ? (hi16.empty() ? &none16 : hi16.data()) : nullptr;
            v.id_base32 = d.id_base16 ? nullptr : (hi32.empty() ? &none32 : hi32.data());
            v.group_id_to_pos = hg.data();
            v.n_groups = d.n_groups;
        }
        secedo::PackedPileup hp_pk;
        const std::string err = secedo::pack_pileup(v, num_cells, max_fragment_length, num_threads,
                                                    block_cells, &secedo::stage_geometry, allow_count_tile,
                                                    &hp_pk);
        if (!err.empty()) {
            h->have_host = h->have_device = false;
            return fail(SECEDO_E_INVALID_ARG, err);
        }
        HIP_TRY(arena_upload(pk.blk_off, hp_pk.blk_off));
        HIP_TRY(arena_upload(pk.entry32, hp_pk.entry32));
        HIP_TRY(arena_upload(pk.mask32, hp_pk.mask32));
        HIP_TRY(arena_upload(pk.entry, hp_pk.entry));
        HIP_TRY(arena_upload(pk.entry_read, hp_pk.entry_read));
        HIP_TRY(arena_upload(pk.range_off, hp_pk.range_off));
        HIP_TRY(arena_upload(pk.read_off, hp_pk.read_off));
        HIP_TRY(arena_upload(pk.read_locus, hp_pk.read_locus));
        HIP_TRY(arena_upload(pk.read_base, hp_pk.read_base));
        pk.num_cells = hp_pk.num_cells;
        pk.block_cells = hp_pk.block_cells;
        pk.num_blocks = hp_pk.num_blocks;
        pk.num_loci = hp_pk.num_loci;
        pk.num_entries = hp_pk.num_entries;
        pk.num_reads = hp_pk.num_reads;
        pk.pair_bound = hp_pk.pair_bound;
        pk.cell_sq = nullptr;
        pk.cell_sq_n = 0;
        pk.cell_sq_host = std::move(hp_pk.cell_sq);
        pk.multi_entries = hp_pk.multi_entries;
        pk.max_read_entries = hp_pk.max_read_entries;
        pk.n_wide = hp_pk.n_wide;
        pk.stage_masks = hp_pk.stage_masks;
        pk.count_tile = hp_pk.count_tile;
        pk.cap_entries = hp_pk.cap_entries;
        pk.cap_loci = hp_pk.cap_loci;
        pk.num_ranges = static_cast<uint32_t>(hp_pk.range_off.size()) - 1;
    }
    h->have_host = h->have_device = false;  // the borrow ends here

    const uint32_t nb = pk.num_blocks;
    if (h->num_tiles != nb * (nb + 1) / 2 || !h->tile_row.p) {
        h->num_tiles = nb * (nb + 1) / 2;
        std::vector<uint16_t> trow, tcol;
        trow.reserve(h->num_tiles);
        tcol.reserve(h->num_tiles);
        for (uint32_t i = 0; i < nb; ++i) {
            for (uint32_t j = i; j < nb; ++j) {
                trow.push_back(static_cast<uint16_t>(i));
                tcol.push_back(static_cast<uint16_t>(j));
            }
        }
        HIP_TRY(h->tile_row.upload(trow));
        HIP_TRY(h->tile_col.upload(tcol));
        h->host_tile_row = trow;
        h->host_tile_col = tcol;
    }
    HIP_TRY(h->counters.ensure((16 + 2048 * 9) * sizeof(unsigned long long)));  // [16..): diagnostic builds
    HIP_TRY(h->max_bits.ensure(sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(h->counters.p, 0, 16 * sizeof(unsigned long long), s));
    h->prepared = true;
    h->timed = false;
    h->flags_ready = false;
    h->wide_known = false;
    return SECEDO_OK;
}

uint32_t secedo_simmat_num_tiles(const secedo_simmat_t *h) { return h ? h->num_tiles : 0; }
uint32_t secedo_simmat_block_cells(const secedo_simmat_t *h) { return h ? h->pk.block_cells : 0; }
uint64_t secedo_simmat_acc_elems(const secedo_simmat_t *h) {
    return h ? static_cast<uint64_t>(h->num_tiles) * h->pk.block_cells * h->pk.block_cells : 0;
}
uint64_t secedo_simmat_num_entries(const secedo_simmat_t *h) { return h ? h->pk.num_entries : 0; }
uint64_t secedo_simmat_num_reads(const secedo_simmat_t *h) { return h ? h->pk.num_reads : 0; }
uint64_t secedo_simmat_num_loci(const secedo_simmat_t *h) { return h ? h->pk.num_loci : 0; }

uint64_t secedo_simmat_pair_bound(const secedo_simmat_t *h) { return h ? h->pk.pair_bound : 0; }
int secedo_simmat_scale_log2(const secedo_simmat_t *h) { return h ? h->scale_log2 : 0; }
uint32_t secedo_simmat_max_read_entries(const secedo_simmat_t *h) { return h ? h->pk.max_read_entries : 0; }
int secedo_simmat_set_scale_bounds(secedo_simmat_t *h, uint64_t pair_bound, uint32_t max_read_entries) {
    if (!h) return fail(SECEDO_E_INVALID_ARG, "handle is null");
    h->pair_bound_override = pair_bound;
    h->max_shared_override = max_read_entries;
    h->override_identity = (pair_bound || max_read_entries) ? h->pileup_identity : 0;
    h->override_dropped = false;
    return SECEDO_OK;
}
int secedo_simmat_scale_bounds_state(const secedo_simmat_t *h) {
    if (!h) return 0;
    if ((h->pair_bound_override || h->max_shared_override) && h->override_identity == h->pileup_identity) return 1;
    return h->override_dropped ? 2 : 0;
}
int secedo_simmat_set_pair_bound(secedo_simmat_t *h, uint64_t pair_bound) {
    return secedo_simmat_set_scale_bounds(h, pair_bound, h ? h->max_shared_override : 0);
}
int secedo_simmat_cell_squares(secedo_simmat_t *h, uint64_t *d_out, void *stream) {
    if (!h || !d_out) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = h->pk.num_cells;
    HIP_TRY(hipMemsetAsync(d_out, 0, n * 8, s));
    if (h->pk.cell_sq && h->pk.cell_sq_n) {
        HIP_TRY(hipMemcpyAsync(d_out, h->pk.cell_sq, std::min<size_t>(n, h->pk.cell_sq_n) * 8, hipMemcpyDeviceToDevice, s));
    } else if (!h->pk.cell_sq_host.empty()) {
        HIP_TRY(hipMemcpyAsync(d_out, h->pk.cell_sq_host.data(), std::min(n, h->pk.cell_sq_host.size()) * 8,
                               hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));  // the source is pageable host memory of the handle
    }
    return SECEDO_OK;
}

int secedo_simmat_zero_acc(secedo_simmat_t *h, int64_t *d_acc, void *stream) {
    if (!h || !d_acc) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(d_acc, 0, secedo_simmat_acc_elems(h) * sizeof(int64_t),
                           static_cast<hipStream_t>(stream)));
    return SECEDO_OK;
}


// The compact lists of the flagged entries (build_flagged_lists), issued on `stream`: the launch stream itself, or
// the handle's side stream -- then behind the fork event and followed by the join event.
struct FlagBuild {
    secedo_simmat *h = nullptr;
    const uint32_t *entry32 = nullptr, *blk_off = nullptr;
    const uint4 *entry = nullptr;
    uint32_t ne = 0;
    size_t n_off = 0;
    hipStream_t stream = nullptr;
    bool done = false;
    static hipError_t run(void *ctx) {
        FlagBuild *b = static_cast<FlagBuild *>(ctx);
        secedo_simmat *h = b->h;
        const bool on_side = h->side.stream && b->stream == h->side.stream;
        hipError_t e = hipSuccess;
        if (on_side && (e = hipStreamWaitEvent(b->stream, h->side.fork, 0)) != hipSuccess) return e;
        e = secedo::build_flagged_lists(b->entry32, b->entry, b->ne, b->blk_off, b->n_off, h->flag_tmp.p, h->flag_tmp.bytes,
                                        h->flag_pre.as<uint32_t>(), h->flag_grp.as<uint32_t>(), h->flag_rec.as<uint4>(),
                                        h->flag_idx.as<uint32_t>(), b->stream);
        if (e == hipSuccess && on_side) e = hipEventRecord(h->side.join, b->stream);
        b->done = true;
        return e;
    }
};

// capacity of the list of read pairs beyond the table per launch (16 bytes each; allocated only for a pileup with a
// read of more than 128 kept entries)
constexpr uint32_t kBeyondCap = 1u << 20;

// tiles [tile_begin, tile_end) when list == nullptr, else the n_list tiles of `list` (global indices)
// overwrite: acc[tiles of the launch] = result instead of +=
static int accumulate_impl(secedo_simmat_t *h, double eps, double hr, double theta, uint32_t tile_begin,
                           uint32_t tile_end, const uint32_t *list, uint32_t n_list, int64_t *d_acc, void *stream,
                           bool overwrite = false, bool *max_done = nullptr) {
    if (max_done) *max_done = false;
    if (!h || !d_acc) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    if (!list && (tile_begin > tile_end || tile_end > h->num_tiles))
        return fail(SECEDO_E_INVALID_ARG, "tile range outside [0, num_tiles]");
    uint64_t list_hash = 0;
    if (list) {
        list_hash = 0xcbf29ce484222325ull;  // FNV-1a over the ids: key of the cached workgroup plan
        for (uint32_t k = 0; k < n_list; ++k) {
            if (list[k] >= h->num_tiles) return fail(SECEDO_E_INVALID_ARG, "tile id outside [0, num_tiles)");
            list_hash = (list_hash ^ list[k]) * 0x100000001b3ull;
        }
        list_hash |= 1ull;
        tile_begin = 0;
        tile_end = n_list;
    }
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool beyond = false;  // this launch notes the read pairs that share more than 128 loci (see below)

    // LLR table: the doubles depend on the rates only, the fixed-point scale on the pileup's pair
    // bound; upload (synchronously, it is rare) only what changed since the last call
    if (!h->have_model || h->lut_eps != eps || h->lut_h != hr || h->lut_theta != theta) {
        h->table = secedo::make_llr_table(eps, hr, theta, h->pk.pair_bound);
        h->lut_eps = eps;
        h->lut_h = hr;
        h->lut_theta = theta;
        h->have_model = true;
        h->have_lut = false;
    }
    // the entries this pileup can reach (no read pair shares more loci than its shorter read has) as the
    // reference evaluates them, wrapping binomial products included (llr_table.hpp)
    // (with shards on several ranks: what ANY of them can reach, so that all quantise the same table)
    const uint32_t reach = std::max(h->max_shared_override, h->pk.max_read_entries);
    if (h->table.ref_upto < std::min(reach, secedo::kLlrRefMax) && !secedo::llr_exact_mode())
        h->have_lut = false;
    if (!secedo::extend_reference(&h->table, reach, std::max(1u, h->num_threads)))
        return fail(SECEDO_E_INVALID_ARG, "these rates give a non-finite log-likelihood ratio (log of 0: the reference "
                                          "would write inf / NaN into the matrix)");
    {
        // the fixed-point scale follows the pair bound of everything that is summed into one accumulator:
        // this pileup's own bound, or the one the caller set for all the shards that will be added up
        const uint64_t bound = std::max(h->pair_bound_override, h->pk.pair_bound);
        // (the bound is a pass over the whole table: kept per table, pair bound and reach)
        if (!h->have_lut || bound != h->scale_for_bound || reach != h->scale_for_reach) {
            h->scale_wanted = secedo::llr_scale_for(h->table, bound, reach);
            h->scale_for_bound = bound;
            h->scale_for_reach = reach;
        }
        const int want_scale = h->scale_wanted;
        if (!h->have_lut || want_scale != h->scale_log2) {
            secedo::requantize(&h->table, want_scale);
            HIP_TRY(h->lut.ensure(h->table.fixed.size() * sizeof(int64_t)));
            // (in the order of the caller's stream, as every upload of accumulate: a launch still running on a
            // non-blocking stream reads the tables and the workgroup plan of ITS call -- a plain hipMemcpy runs on the
            // null stream, which such streams do not wait for; the host sources stay alive in the handle)
            HIP_TRY(h->uploads.put(h->lut.p, h->table.fixed.data(), h->table.fixed.size() * sizeof(int64_t), s));
            h->scale_log2 = want_scale;
            h->model = h->table.model;
            h->have_lut = true;
        }
        secedo::SlowPathArgs sp;
        std::memset(&sp, 0, sizeof(sp));
        sp.entry = h->pk.entry.as<uint4>();
        sp.entry_read = h->pk.entry_read.as<uint32_t>();
        sp.read_off = h->pk.read_off.as<uint32_t>();
        sp.read_locus = h->pk.read_locus.as<uint32_t>();
        sp.read_base = h->pk.read_base.as<uint8_t>();
        sp.lut = h->lut.as<long long>();
        const secedo::LlrModel &m = h->table.model;
        sp.model = secedo::LlrModelDev{m.ln_u1, m.ln_v1, m.ln_u2, m.ln_v2, m.ln_w1, m.ln_z1, m.ln_w2, m.ln_z2};
        sp.scale_log2 = h->scale_log2;
        // Read pairs that share more than 128 loci (possible when a read has more kept entries than that): what the
        // reference returns there is the value of its wrapped integer sums -- O(x_s^2 x_d^2) terms per entry, nothing
        // for a table. The kernels note such pairs instead of adding their joint term, and after the launch the
        // host evaluates the few distinct (x_s, x_d) as the reference does and adds them (below). SECEDO_LLR_EXACT=1:
        // the closed form on the device, nothing noted.
        if (reach > secedo::kLlrRefMax && !secedo::llr_exact_mode()) {
            HIP_TRY(h->beyond_list.ensure((size_t)kBeyondCap * sizeof(uint4)));
            HIP_TRY(h->beyond_count.ensure(sizeof(uint32_t)));
            sp.beyond_list = h->beyond_list.as<uint4>();
            sp.beyond_count = h->beyond_count.as<uint32_t>();
            sp.beyond_cap = kBeyondCap;
            beyond = true;
        }
        if (!h->have_slow || std::memcmp(&sp, &h->slow_host, sizeof(sp)) != 0) {
            HIP_TRY(h->slow_args.ensure(sizeof(sp)));
            h->slow_host = sp;
            HIP_TRY(h->uploads.put(h->slow_args.p, &sp, sizeof(sp), s));
            h->have_slow = true;
        }
    }

    if (h->pk.num_entries == 0) {
        // nothing to add (a rank whose shard is empty): the table and the scale above are all finalize needs
        if (overwrite) {  // ... and the tiles of the launch are zero
            const size_t b2 = (size_t)h->pk.block_cells * h->pk.block_cells;
            if (list) {
                for (uint32_t k = 0; k < n_list; ++k)
                    HIP_TRY(hipMemsetAsync(d_acc + (size_t)list[k] * b2, 0, b2 * sizeof(int64_t), s));
            } else {
                HIP_TRY(hipMemsetAsync(d_acc + (size_t)tile_begin * b2, 0, (size_t)(tile_end - tile_begin) * b2 * sizeof(int64_t), s));
            }
        }
        HIP_TRY(hipMemsetAsync(h->counters.p, 0, 96 * sizeof(unsigned long long), s));
        HIP_TRY(hipEventRecord(h->ev_begin, s));
        HIP_TRY(hipEventRecord(h->ev_end, s));
        h->timed = true;
        h->timed_mid = false;  // no pair kernel ran: last_pair_kernel_ms has nothing to report
        return SECEDO_OK;
    }
    const uint32_t n_tiles = tile_end - tile_begin;
    secedo::AccumulateArgs a;
    a.blk_off = h->pk.blk_off.as<uint32_t>();
    a.stride = h->pk.num_loci + 1;
    a.entry32 = h->pk.entry32.as<uint32_t>();
    a.mask32 = h->pk.mask32.as<uint32_t>();
    a.entry = h->pk.entry.as<uint4>();
    a.range_off = h->pk.range_off.as<uint32_t>();
    a.num_ranges = h->pk.num_ranges;
    a.tile_row = h->tile_row.as<uint16_t>();
    a.tile_col = h->tile_col.as<uint16_t>();
    a.tile_begin = tile_begin;
    a.tile_ids = nullptr;
    if (list) {
        if (h->plan_list_hash != list_hash) {
            HIP_TRY(h->tile_ids.ensure((size_t)std::max(n_list, 1u) * 4));
            HIP_TRY(h->uploads.put(h->tile_ids.p, list, (size_t)n_list * 4, s));
        }
        a.tile_ids = h->tile_ids.as<uint32_t>();
    }
    // Workgroups: every tile is cut into chunks of locus ranges so that the launch fills the 256 CUs
    // in whole rounds of about equally loaded workgroups (a diagonal tile holds half the pairs of an
    // off-diagonal one and gets half the chunks). Cached per tile range.
    if (h->plan_tile_begin != tile_begin || h->plan_tile_end != tile_end || h->plan_ranges != h->pk.num_ranges
        || h->plan_blocks != h->pk.num_blocks || h->plan_list_hash != list_hash) {
        // workgroups resident per CU (LDS-limited): 1 (128-cell tiles), 2 (64-cell tiles with staged masks,
        // 512 threads), 4 (the other 64-cell variants)
        // (the 64-cell count tile runs accumulate_counts with 512 threads and 59 KiB of LDS: two per CU)
        const uint32_t wgs_per_round = h->pk.block_cells == 128 ? 256u : (h->pk.stage_masks || h->pk.count_tile) ? 512u : 1024u;
        uint32_t rounds = 1;
        if (const char *env = std::getenv("SECEDO_ROUNDS")) rounds = std::max(1, std::atoi(env));
        // weight of a tile = the time it takes: per row-side entry a fixed part (the batch set-up) and a
        // part per column entry of the same locus (the pairs; half of them in a diagonal tile). Fitted to
        // the per-workgroup times on C2: the fixed part is worth 5.7 pairs.
        const uint32_t Bc = h->pk.block_cells;
        auto cells_of = [&](uint32_t blk) { return (double)std::min(Bc, h->pk.num_cells - blk * Bc); };
        const double per_cell_locus = h->pk.num_loci ? (double)h->pk.num_entries / h->pk.num_cells / h->pk.num_loci : 0.0;
        std::vector<double> weight(n_tiles);
        double total_weight = 0;
        for (uint32_t k = 0; k < n_tiles; ++k) {
            const uint32_t t = list ? list[k] : tile_begin + k;
            const uint32_t I = h->host_tile_row[t], J = h->host_tile_col[t];
            const double depth = per_cell_locus * cells_of(J) * (I == J ? 0.5 : 1.0);  // column entries per locus
            weight[k] = cells_of(I) * (5.7 + depth);
            total_weight += weight[k];
        }
        // whole rounds of workgroups, shared out so that the slowest chunk is as fast as possible: every
        // tile starts with one chunk and the next one always goes to the tile whose chunks are heaviest
        const uint64_t slots = static_cast<uint64_t>(wgs_per_round) * std::max<uint64_t>(rounds, (n_tiles + wgs_per_round - 1) / wgs_per_round);
        const uint32_t max_chunks = std::max(1u, h->pk.num_ranges);
        std::vector<uint32_t> chunks(n_tiles, 1u);
        if (total_weight > 0 && n_tiles < slots && n_tiles <= 2 * wgs_per_round) {  // more tiles: one workgroup each
            std::priority_queue<std::pair<double, uint32_t>> heaviest;  // (weight per chunk, tile)
            for (uint32_t k = 0; k < n_tiles; ++k) heaviest.push({weight[k], k});
            for (uint64_t given = n_tiles; given < slots && !heaviest.empty();) {
                const uint32_t k = heaviest.top().second;
                heaviest.pop();
                if (chunks[k] >= max_chunks) continue;  // one range per chunk at least
                ++chunks[k];
                ++given;
                heaviest.push({weight[k] / chunks[k], k});
            }
        }
        std::vector<uint32_t> wg_begin(n_tiles + 1, 0);
        std::vector<uint32_t> wg_tile;
        for (uint32_t t = 0; t < n_tiles; ++t) {
            wg_begin[t + 1] = wg_begin[t] + chunks[t];
            for (uint32_t k = 0; k < chunks[t]; ++k) wg_tile.push_back(t);
        }
        // (a larger plan than any before: the buffers grow, and hipFree waits for the device; otherwise the new plan
        // follows the launches that read the old one in stream order)
        HIP_TRY(h->plan_wg_tile.ensure(wg_tile.size() * 4));
        HIP_TRY(h->plan_wg_begin.ensure(wg_begin.size() * 4));
        HIP_TRY(h->uploads.put(h->plan_wg_tile.p, wg_tile.data(), wg_tile.size() * 4, s));
        HIP_TRY(h->uploads.put(h->plan_wg_begin.p, wg_begin.data(), wg_begin.size() * 4, s));
        h->plan_workgroups = wg_begin[n_tiles];
        h->plan_tile_begin = tile_begin;
        h->plan_tile_end = tile_end;
        h->plan_list_hash = list_hash;
        h->plan_ranges = h->pk.num_ranges;
        h->plan_blocks = h->pk.num_blocks;
    }
    a.n_tiles = n_tiles;
    a.n_workgroups = h->plan_workgroups;
    a.wg_tile = h->plan_wg_tile.as<uint32_t>();
    a.tile_wg_begin = h->plan_wg_begin.as<uint32_t>();
    a.debug = 0;
    if (const char *env = std::getenv("SECEDO_DEBUG_ABLATE")) a.debug = static_cast<uint32_t>(std::atoi(env));
    a.lut = h->lut.as<long long>();
    a.slow = h->slow_args.as<secedo::SlowPathArgs>();
    a.acc = d_acc;
    a.overwrite = overwrite;
    // the maximum finalize needs, on the way (assign_finalize: all tiles, stored, one workgroup per tile)
    if (max_done && !beyond && overwrite && !list && tile_begin == 0 && tile_end == h->num_tiles && h->pk.count_tile
        && !h->pk.stage_masks && secedo::counts_path_enabled() && secedo::counts_split(n_tiles) == 1) {
        HIP_TRY(hipMemsetAsync(h->max_bits.p, 0, sizeof(unsigned long long), s));
        a.max_bits = h->max_bits.as<unsigned long long>();
        a.max_scale = std::ldexp(1.0, -h->scale_log2);
        *max_done = true;
    }
    a.counters = h->counters.as<unsigned long long>();
    if (h->pk.stage_masks && h->pk.block_cells == 64) {
        // accumulate_masks pairs from the 8-locus windows alone; the entries of reads that reach beyond them are
        // listed per cell block once per prepare for its second kernel (SECEDO_MASKS_KERNEL=0: accumulate_tiles)
        static const bool allowed_env = [] { const char *e = std::getenv("SECEDO_MASKS_KERNEL"); return !(e && std::atoi(e) == 0); }();
        // (the list's scan is one workgroup: beyond 1024 cell blocks -- more than num_cells allows today -- the pileup
        // keeps to accumulate_tiles instead of failing the call)
        const bool allowed = allowed_env && h->pk.num_blocks <= 1024u;
        if (allowed && !h->wide_known) {
            // how many there are came with the packing's last read-back (DevicePacked::n_wide): count per block, scan
            // and fill are enqueued behind each other, nothing waits for the device here
            const uint32_t nb = h->pk.num_blocks;
            h->n_wide = h->pk.n_wide;
            if (h->n_wide) {
                HIP_TRY(h->wide_tab.ensure(((size_t)3 * nb + 2) * 4));
                uint32_t *cnt = h->wide_tab.as<uint32_t>(), *off = cnt + nb, *cur = off + nb + 1;
                HIP_TRY(secedo::wide_count(a.entry32, a.blk_off, a.stride, nb, cnt, off, cur, s));
                HIP_TRY(h->wide_list.ensure((size_t)h->n_wide * 4));
                HIP_TRY(secedo::wide_fill(a.entry32, a.blk_off, a.stride, nb, cur, h->wide_list.as<uint32_t>(), s));
            }
            // ... and the words the kernel pairs from, once per prepare
            const size_t ne = (size_t)h->pk.num_entries;
            HIP_TRY(h->mk_words.ensure(std::max<size_t>(ne, 1) * 12));
            HIP_TRY(secedo::masks_words(a.entry32, a.mask32, (uint32_t)ne, h->mk_words.as<uint32_t>(),
                                        h->mk_words.as<uint32_t>() + ne, h->mk_words.as<uint32_t>() + 2 * ne, s));
            h->wide_known = true;
        }
        if (allowed) {
            const size_t ne = (size_t)h->pk.num_entries;
            a.mk_y = h->mk_words.as<uint32_t>();
            a.mk_xcol = a.mk_y + ne;
            a.mk_xrow = a.mk_y + 2 * ne;
        }
        a.masks_kernel = allowed;
        if (allowed && h->n_wide) {
            a.wide_off = h->wide_tab.as<uint32_t>() + h->pk.num_blocks;
            a.wide_list = h->wide_list.as<uint32_t>();
        }
    }

    HIP_TRY(hipMemsetAsync(h->counters.p, 0, 96 * sizeof(unsigned long long), s));
    if (beyond) HIP_TRY(hipMemsetAsync(h->beyond_count.p, 0, sizeof(uint32_t), s));
    const secedo::SideStream *side = nullptr;
    secedo::SideStream side_call;
    FlagBuild build;
    if (h->pk.count_tile && !h->pk.stage_masks && secedo::counts_path_enabled()) {
        // accumulate_counts + correct_tiles. The compact list of flagged entries, once per prepare, is read by
        // the second kernel only: it is built on a stream of the handle's own while the pair kernel runs
        static const bool serial = [] {
            const char *e = std::getenv("SECEDO_CORRECT_SERIAL");
            return e && std::atoi(e) != 0;
        }();
        if (!serial) {
            if (!h->side.stream) {
                HIP_TRY(hipStreamCreateWithFlags(&h->side.stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&h->side.fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&h->side.join, hipEventDisableTiming));
            }
            side = &h->side;
        }
        if (!h->flags_ready) {
            const uint32_t ne = (uint32_t)h->pk.num_entries;
            const size_t scan_bytes = secedo::flagged_scan_bytes(ne);
            HIP_TRY(h->flag_tmp.ensure(std::max<size_t>(scan_bytes, 16)));
            HIP_TRY(h->flag_pre.ensure(((size_t)ne + 1) * 4));
            HIP_TRY(h->flag_rec.ensure(std::max<size_t>(ne, 1) * 16));
            HIP_TRY(h->flag_idx.ensure(std::max<size_t>(ne, 1) * 4));
            const size_t n_off = (size_t)h->pk.num_blocks * a.stride;
            HIP_TRY(h->flag_grp.ensure(std::max<size_t>(n_off, 1) * 4));
            build.h = h;
            build.entry32 = a.entry32;
            build.entry = a.entry;
            build.blk_off = a.blk_off;
            build.ne = ne;
            build.n_off = n_off;
            if (side) {
                // the lists depend on the packed pileup (complete on `s` by now) and are read by correct_tiles
                // only: the host enqueues their kernels behind the pair kernel's launch, on the side stream
                HIP_TRY(hipEventRecord(side->fork, s));
                build.stream = side->stream;
                side_call = *side;
                side_call.deferred = &FlagBuild::run;
                side_call.deferred_ctx = &build;
                side = &side_call;
            } else {
                build.stream = s;
                if (FlagBuild::run(&build) != hipSuccess) return fail(SECEDO_E_HIP, "building the flagged entries' lists failed");
            }
            // (flags_ready is set once the lists' kernels have been ISSUED without an error, below: a launch that
            // fails in between must not leave the next accumulate reading lists nobody built -- ADVICE r03)
        }
        a.flag_grp = h->flag_grp.as<uint32_t>();
        a.flag_rec = h->flag_rec.as<uint4>();
        a.flag_idx = h->flag_idx.as<uint32_t>();
        {
            const double per_block_locus = h->pk.num_loci && h->pk.num_blocks
                    ? (double)h->pk.num_entries / h->pk.num_loci / h->pk.num_blocks : 0.0;
            a.group_hint = per_block_locus < 2.5 ? 2 : per_block_locus < 3.2 ? 3 : 4;
        }
    }
    HIP_TRY(hipEventRecord(h->ev_begin, s));
    // 16-bit pair counters per cell pair are safe when no cell pair can collect 65536 pairs
    const bool count_tile = h->pk.count_tile;
    HIP_TRY(h->slab.ensure(secedo::accumulate_slab_bytes(h->pk.block_cells, count_tile, a.n_workgroups)));
    a.slab = h->slab.p;
    h->timed_mid = count_tile && !h->pk.stage_masks && secedo::counts_path_enabled();
    HIP_TRY(secedo::launch_accumulate(a, h->pk.block_cells, h->pk.stage_masks, count_tile, n_tiles, s, side,
                                      h->timed_mid ? h->ev_mid : nullptr));
    if (build.h && build.stream != s && !build.done) {  // no tile in the launch: nobody issued the lists yet
        if (FlagBuild::run(&build) != hipSuccess) return fail(SECEDO_E_HIP, "building the flagged entries' lists failed");
        HIP_TRY(hipStreamWaitEvent(s, h->side.join, 0));
    }
    if (build.h && build.done) h->flags_ready = true;
    if (beyond) {
        // the pairs the kernels noted: the call waits for the launch here (only a pileup with a read of more than 128
        // kept entries comes this way), evaluates the distinct (x_s, x_d) as the reference does and adds the terms
        uint32_t n_noted = 0;
        HIP_TRY(hipMemcpyAsync(&n_noted, h->beyond_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (n_noted > kBeyondCap)
            return fail(SECEDO_E_LIMIT, std::to_string(n_noted) + " read pairs of this launch share more than 128 loci (at most "
                                        + std::to_string(kBeyondCap) + " per launch: accumulate fewer tiles at a time, or set "
                                        "SECEDO_LLR_EXACT=1 for the formula in exact arithmetic)");
        if (n_noted) {
            std::vector<uint32_t> noted((size_t)n_noted * 4);
            HIP_TRY(hipMemcpy(noted.data(), h->beyond_list.p, noted.size() * 4, hipMemcpyDeviceToHost));
            const uint32_t Bc = h->pk.block_cells, nb = h->pk.num_blocks;
            std::vector<unsigned long long> index(n_noted);
            std::vector<long long> value(n_noted);
            for (uint32_t k = 0; k < n_noted; ++k) {
                uint32_t ca = noted[(size_t)k * 4], cb = noted[(size_t)k * 4 + 1];
                const uint32_t xs = noted[(size_t)k * 4 + 2], xd = noted[(size_t)k * 4 + 3];
                if (ca / Bc > cb / Bc) std::swap(ca, cb);  // the tile's row block is the smaller one
                const uint32_t I = ca / Bc, J = cb / Bc;
                const uint64_t tile = (uint64_t)I * nb - (uint64_t)I * (I - 1) / 2 + (J - I);  // row-major upper triangle
                if (ca >= h->pk.num_cells || cb >= h->pk.num_cells || tile >= h->num_tiles || h->host_tile_row[tile] != I
                    || h->host_tile_col[tile] != J)
                    return fail(SECEDO_E_STATE, "a noted read pair lies outside the matrix");
                const double d = secedo::reference_llr_any(eps, hr, theta, xs, xd, std::max(1u, h->num_threads));
                if (!std::isfinite(d) || std::fabs(std::ldexp(d, h->scale_log2)) >= 0x1p62)
                    return fail(SECEDO_E_INVALID_ARG, "a read pair sharing " + std::to_string(xs + xd) + " loci has a non-finite "
                                                      "log-likelihood ratio in the reference's arithmetic (it would write "
                                                      "inf / NaN into the matrix); SECEDO_LLR_EXACT=1 selects the exact formula");
                index[k] = tile * Bc * Bc + (uint64_t)(ca % Bc) * Bc + (cb % Bc);
                value[k] = std::llround(std::ldexp(d, h->scale_log2));
            }
            HIP_TRY(h->beyond_index.ensure((size_t)n_noted * 8));
            HIP_TRY(h->beyond_value.ensure((size_t)n_noted * 8));
            HIP_TRY(hipMemcpyAsync(h->beyond_index.p, index.data(), (size_t)n_noted * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(h->beyond_value.p, value.data(), (size_t)n_noted * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(secedo::launch_add_terms(d_acc, h->beyond_index.as<unsigned long long>(), h->beyond_value.as<long long>(),
                                             n_noted, s));
            HIP_TRY(hipStreamSynchronize(s));  // (the sources are this frame's vectors)
        }
    }
    h->timed_mid = h->timed_mid && n_tiles > 0;
    HIP_TRY(hipEventRecord(h->ev_end, s));
    h->timed = true;
    return SECEDO_OK;
}

int secedo_simmat_accumulate(secedo_simmat_t *h, double eps, double hr, double theta, uint32_t tile_begin,
                             uint32_t tile_end, int64_t *d_acc, void *stream) {
    return accumulate_impl(h, eps, hr, theta, tile_begin, tile_end, nullptr, 0, d_acc, stream);
}

int secedo_simmat_assign(secedo_simmat_t *h, double eps, double hr, double theta, uint32_t tile_begin,
                         uint32_t tile_end, int64_t *d_acc, void *stream) {
    return accumulate_impl(h, eps, hr, theta, tile_begin, tile_end, nullptr, 0, d_acc, stream, true);
}

int secedo_simmat_assign_list(secedo_simmat_t *h, double eps, double hr, double theta, const uint32_t *tile_ids,
                              uint32_t n_tile_ids, int64_t *d_acc, void *stream) {
    if (!tile_ids && n_tile_ids) return fail(SECEDO_E_INVALID_ARG, "tile_ids is null");
    static const uint32_t none = 0;
    return accumulate_impl(h, eps, hr, theta, 0, 0, tile_ids ? tile_ids : &none, n_tile_ids, d_acc, stream, true);
}

int secedo_simmat_accumulate_list(secedo_simmat_t *h, double eps, double hr, double theta, const uint32_t *tile_ids,
                                  uint32_t n_tile_ids, int64_t *d_acc, void *stream) {
    if (!tile_ids && n_tile_ids) return fail(SECEDO_E_INVALID_ARG, "tile_ids is null");
    static const uint32_t none = 0;
    return accumulate_impl(h, eps, hr, theta, 0, 0, tile_ids ? tile_ids : &none, n_tile_ids, d_acc, stream);
}

int secedo_simmat_tiles_of_rows(const secedo_simmat_t *h, uint32_t row_begin, uint32_t row_end, uint32_t *tile_ids,
                                uint32_t *n_tile_ids) {
    if (!h || !n_tile_ids) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    if (row_begin > row_end || row_end > h->pk.num_cells) return fail(SECEDO_E_INVALID_ARG, "row range outside the matrix");
    uint32_t n = 0;
    if (row_begin < row_end) {
        const uint32_t B = h->pk.block_cells, b0 = row_begin / B, b1 = (row_end - 1) / B;
        for (uint32_t t = 0; t < h->num_tiles; ++t) {
            const uint32_t I = h->host_tile_row[t], J = h->host_tile_col[t];
            if ((I >= b0 && I <= b1) || (J >= b0 && J <= b1)) {
                if (tile_ids) tile_ids[n] = t;
                ++n;
            }
        }
    }
    *n_tile_ids = n;
    return SECEDO_OK;
}

int secedo_simmat_max_of_tiles(secedo_simmat_t *h, const int64_t *d_acc, const uint32_t *tile_ids, uint32_t n_tile_ids,
                               double *max_value, void *stream) {
    if (!h || !d_acc || !max_value || (!tile_ids && n_tile_ids)) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    if (!h->have_lut) return fail(SECEDO_E_STATE, "accumulate was not called");
    for (uint32_t k = 0; k < n_tile_ids; ++k)
        if (tile_ids[k] >= h->num_tiles) return fail(SECEDO_E_INVALID_ARG, "tile id outside [0, num_tiles)");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    DevBuf ids;
    HIP_TRY(ids.ensure((size_t)std::max(n_tile_ids, 1u) * 4));
    if (n_tile_ids) HIP_TRY(hipMemcpyAsync(ids.p, tile_ids, (size_t)n_tile_ids * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(secedo::launch_tile_max(d_acc, h->tile_row.as<uint16_t>(), h->tile_col.as<uint16_t>(), ids.as<uint32_t>(),
                                    n_tile_ids, h->pk.block_cells, h->scale_log2, h->max_bits.as<unsigned long long>(),
                                    s));
    unsigned long long bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, h->max_bits.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(max_value, &bits, 8);
    return SECEDO_OK;
}

int secedo_simmat_finalize_rows_max(secedo_simmat_t *h, int normalization, const int64_t *d_acc, uint32_t row_begin,
                                    uint32_t row_end, double max_value, double *d_out_rows, void *stream) {
    if (normalization < 0 || normalization > 2)
        return fail(SECEDO_E_INVALID_NORMALIZATION, "Invalid normalization: " + std::to_string(normalization));
    if (!h || !d_acc || !d_out_rows) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    if (!h->have_lut) return fail(SECEDO_E_STATE, "accumulate was not called");
    if (row_begin > row_end || row_end > h->pk.num_cells) return fail(SECEDO_E_INVALID_ARG, "row range outside the matrix");
    if (!(max_value >= 0.0)) return fail(SECEDO_E_INVALID_ARG, "max_value must be the (non-negative) maximum of D");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long bits;
    std::memcpy(&bits, &max_value, 8);
    HIP_TRY(hipMemcpyAsync(h->max_bits.p, &bits, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // `bits` lives on this stack frame
    HIP_TRY(secedo::launch_finalize(d_acc, h->tile_row.as<uint16_t>(), h->tile_col.as<uint16_t>(), h->num_tiles,
                                    h->pk.num_cells, h->pk.block_cells, h->scale_log2, normalization,
                                    h->max_bits.as<unsigned long long>(), row_begin, row_end, d_out_rows, s, true));
    return SECEDO_OK;
}

static int finalize_mode(secedo_simmat_t *h, int mode, const int64_t *d_acc, uint32_t row_begin, uint32_t row_end,
                         double *d_out, void *stream, bool keep_max = false) {
    if (!h || !d_acc || !d_out) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    if (!h->have_lut) return fail(SECEDO_E_STATE, "accumulate was not called");
    if (row_begin > row_end || row_end > h->pk.num_cells) return fail(SECEDO_E_INVALID_ARG, "row range outside the matrix");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(secedo::launch_finalize(d_acc, h->tile_row.as<uint16_t>(), h->tile_col.as<uint16_t>(), h->num_tiles,
                                    h->pk.num_cells, h->pk.block_cells, h->scale_log2, mode,
                                    h->max_bits.as<unsigned long long>(), row_begin, row_end, d_out,
                                    static_cast<hipStream_t>(stream), keep_max));
    return SECEDO_OK;
}

int secedo_simmat_assign_finalize(secedo_simmat_t *h, double eps, double hr, double theta, int normalization,
                                  int64_t *d_acc, double *d_out, void *stream) {
    if (normalization < 0 || normalization > 2)
        return fail(SECEDO_E_INVALID_NORMALIZATION, "Invalid normalization: " + std::to_string(normalization));
    if (!h || !d_acc || !d_out) return fail(SECEDO_E_INVALID_ARG, "null argument");
    bool max_done = false;
    const int rc = accumulate_impl(h, eps, hr, theta, 0, h->prepared ? h->num_tiles : 0, nullptr, 0, d_acc, stream, true,
                                   (normalization == 0 || normalization == 2) ? &max_done : nullptr);
    if (rc != SECEDO_OK) return rc;
    return finalize_mode(h, normalization, d_acc, 0, h->pk.num_cells, d_out, stream, max_done);
}

int secedo_simmat_finalize(secedo_simmat_t *h, int normalization, const int64_t *d_acc, double *d_out,
                           void *stream) {
    if (normalization < 0 || normalization > 2)
        return fail(SECEDO_E_INVALID_NORMALIZATION, "Invalid normalization: " + std::to_string(normalization));
    return finalize_mode(h, normalization, d_acc, 0, h ? h->pk.num_cells : 0, d_out, stream);
}

int secedo_simmat_finalize_rows(secedo_simmat_t *h, int normalization, const int64_t *d_acc, uint32_t row_begin,
                                uint32_t row_end, double *d_out_rows, void *stream) {
    if (normalization < 0 || normalization > 2)
        return fail(SECEDO_E_INVALID_NORMALIZATION, "Invalid normalization: " + std::to_string(normalization));
    return finalize_mode(h, normalization, d_acc, row_begin, row_end, d_out_rows, stream);
}

int secedo_simmat_finalize_raw(secedo_simmat_t *h, const int64_t *d_acc, double *d_out, void *stream) {
    return finalize_mode(h, 3, d_acc, 0, h ? h->pk.num_cells : 0, d_out, stream);
}

int secedo_simmat_last_counts(secedo_simmat_t *h, uint64_t *updates, uint64_t *read_pairs) {
    if (!h) return fail(SECEDO_E_INVALID_ARG, "handle is null");
    if (!h->prepared) return fail(SECEDO_E_STATE, "prepare was not called");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long c[2] = {0, 0};
    HIP_TRY(hipMemcpy(c, h->counters.p, sizeof(c), hipMemcpyDeviceToHost));
    if (updates) *updates = c[0];
    if (read_pairs) *read_pairs = c[1];
    if (std::getenv("SECEDO_STAMPS_PRINT")) {  // diagnostic builds (-DSECEDO_STAMPS) only
        unsigned long long st[16] = {0};
        HIP_TRY(hipMemcpy(st, h->counters.p, sizeof(st), hipMemcpyDeviceToHost));
        if (std::getenv("SECEDO_STAMPS_COUNTS")) {  // correct_tiles (-DSECEDO_STAMPS)
            unsigned long long cf[11] = {0};
            HIP_TRY(hipMemcpy(cf, h->counters.as<unsigned long long>() + 82, sizeof(cf), hipMemcpyDeviceToHost));
            if (cf[0])
                std::fprintf(stderr, "[stamps-correct] pair tests %llu, tail terms %llu, joint terms %llu, later-locus pairs %llu | "
                                     "per wave: pairs phase %.0f cycles (%.0f until the first q records are there), flush phase %.0f cycles, "
                                     "flagged row entries %.0f; busiest lane: %.1f q iterations, %.0f cycles until their records are there\n",
                             cf[0], cf[1], cf[2], cf[3], (double)cf[4] / (double)cf[6], (double)cf[8] / (double)cf[6],
                             (double)cf[5] / (double)cf[6], (double)cf[7] / (double)cf[6], (double)cf[10] / (double)cf[6],
                             (double)cf[9] / (double)cf[6]);
        }
        if (st[8] && std::getenv("SECEDO_STAMPS_COUNTS")) {  // accumulate_counts (-DSECEDO_STAMPS)
            const double w = (double)st[8];
            std::fprintf(stderr, "[stamps-counts] sampled waves %llu | per wave cycles: total %.0f barrier-A %.0f stage+barrier-B %.0f "
                                 "items(+col prefetch issue) %.0f primary groups %.0f drain %.0f | per primary batch %.0f, per drain batch %.0f "
                                 "(%.2f drain batches per primary batch)\n",
                         st[8], st[7] / w, st[2] / w, st[3] / w, st[4] / w, st[5] / w, st[6] / w,
                         (double)st[5] / std::max<double>(1, (double)st[9]), (double)st[6] / std::max<double>(1, (double)st[10]),
                         (double)st[10] / std::max<double>(1, (double)st[9]));
            unsigned long long pw[64] = {0};
            HIP_TRY(hipMemcpy(pw, h->counters.as<unsigned long long>() + 16, sizeof(pw), hipMemcpyDeviceToHost));
            const double wgs = w / 16.0;
            for (int k = 0; k < 16; ++k)
                std::fprintf(stderr, "[stamps-counts] wave %2d: barrier-A %8.0f  pairs %8.0f  items %8.0f  stage+B %8.0f\n", k,
                             pw[k * 4] / wgs, pw[k * 4 + 1] / wgs, pw[k * 4 + 2] / wgs, pw[k * 4 + 3] / wgs);
        } else if (st[8]) {
            std::fprintf(stderr, "[stamps] waves %llu batches %llu trips %llu | per wave: lifetime %.0f cyc, setup %.0f, "
                                 "fill %.0f, trips %.0f | per batch: setup %.0f fill %.0f trips %.0f (%.2f trips)\n",
                         st[8], st[5], st[6], (double)st[7] / st[8], (double)st[2] / st[8], (double)st[3] / st[8],
                         (double)st[4] / st[8], (double)st[2] / st[5], (double)st[3] / st[5], (double)st[4] / st[5],
                         (double)st[6] / st[5]);
            std::fprintf(stderr, "[stamps] per wave: barrier-1 wait %.0f, staging+barrier-2 %.0f, prefetch issue %.0f\n",
                         (double)st[9] / st[8], (double)st[10] / st[8], (double)st[11] / st[8]);
            std::fprintf(stderr, "[stamps] per wave: post-trip %.0f, batch loop total %.0f\n",
                         (double)st[12] / st[8], (double)st[13] / st[8]);
            std::fprintf(stderr, "[stamps] longest wave lifetime %llu cyc; ranges %u workgroups %u\n", st[14],
                         h->pk.num_ranges, h->plan_workgroups);
            std::vector<unsigned long long> wg(std::min<uint32_t>(h->plan_workgroups, 2048u));
            HIP_TRY(hipMemcpy(wg.data(), h->counters.as<unsigned long long>() + 16, wg.size() * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> det(wg.size() * 8);
            HIP_TRY(hipMemcpy(det.data(), h->counters.as<unsigned long long>() + 16 + 2048, det.size() * 8, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < wg.size(); ++k)
                std::fprintf(stderr, "[stamps-wg] %zu %llu %llu %llu | hwid %llx ranges %llu listflush %llu endbarrier %llu slab %llu | realbegin %llu realend %llu batchloop %llu\n", k,
                             wg[k] & 0xFFFFFFFFull, (wg[k] >> 32) & 0xFFF, (wg[k] >> 44) & 0xFFF, det[k * 8], det[k * 8 + 1] / 1000,
                             det[k * 8 + 2] / 1000, det[k * 8 + 3] / 1000, det[k * 8 + 4] / 1000, det[k * 8 + 5], det[k * 8 + 6], det[k * 8 + 7] / 1000);
        }
    }
    return SECEDO_OK;
}

int secedo_simmat_last_accumulate_ms(secedo_simmat_t *h, float *ms) {
    if (!h || !ms) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->timed) return fail(SECEDO_E_STATE, "accumulate was not called");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(h->ev_end));
    HIP_TRY(hipEventElapsedTime(ms, h->ev_begin, h->ev_end));
    return SECEDO_OK;
}

const char *secedo_simmat_pair_kernel(const secedo_simmat_t *h) {
    if (!h || !h->prepared) return "";
    if (h->pk.count_tile && !h->pk.stage_masks && secedo::counts_path_enabled()) return "accumulate_counts";
    if (h->pk.stage_masks && h->pk.block_cells == 64 && h->pk.num_blocks <= 1024u) {
        const char *e = std::getenv("SECEDO_MASKS_KERNEL");
        if (!(e && std::atoi(e) == 0)) return "accumulate_masks";
    }
    return "accumulate_tiles";
}

int secedo_simmat_last_pair_kernel_ms(secedo_simmat_t *h, float *ms) {
    if (!h || !ms) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (!h->timed || !h->timed_mid) return fail(SECEDO_E_STATE, "the last accumulate did not run the sparse-loci kernels");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(h->ev_mid));
    HIP_TRY(hipEventElapsedTime(ms, h->ev_begin, h->ev_mid));
    return SECEDO_OK;
}


// ------------------------------------------------------------------------------------------------
// Several GPUs behind the one-shot call (north_star: "the N x N output is block-partitioned across the 8 GPUs of
// one node"; SURVEY.md 8b set_devices, section 5 SECEDO_GPUS). The reference's caller (spectral_clustering.cpp:
// 354-356) keeps calling computeSimilarityMatrix() with the same signature; ONE process drives N devices, a host
// thread ("lane") per device:
//   1. every lane uploads the flat pileup to its device and packs it there (the packing is replicated, the
//      lanes run side by side: nothing travels between devices before the accumulators exist);
//   2. lane k stores its contiguous range of upper-triangular tiles into its own tile-major int64 accumulator,
//      in kChunks launches with an event behind each;
//   3. every lane pulls the other lanes' tiles into its accumulator chunk by chunk as their events fire
//      (hipMemcpyPeerAsync over xGMI on a copy stream of its own: the all-gather of SURVEY 8e as N x (N - 1)
//      direct copies, which point-to-point links serve better than a ring), so that the exchange of a lane's
//      first chunks runs behind the accumulation of its later ones;
//   4. every lane normalises ITS block of rows from the complete accumulator (the maximum ADD_MIN / SCALE_MAX_1
//      need is taken over all tiles on every device: no scalar exchange) and downloads it straight into its
//      rows of the caller's matrix -- N downloads over N PCIe links instead of one.
// Integer accumulators make the result bit-identical to the single-device call whatever N is; a device may be
// listed more than once (that is how the one-GPU test box runs this code: tests/test_gpu_multi_device.py).
// ------------------------------------------------------------------------------------------------
namespace {

std::mutex g_devices_mutex;
std::vector<int> g_devices;     // secedo_simmat_set_devices; empty: SECEDO_GPUS, else SECEDO_DEVICE, else 0
std::mutex g_multi_mutex;       // one multi-device call at a time (the lanes' rings and pool slots are per lane)

int parse_device_list(const char *text, std::vector<int> *out) {
    out->clear();
    const std::string t(text);
    if (t.find(',') == std::string::npos) {  // a count: devices 0 .. N - 1
        char *end = nullptr;
        const long n = std::strtol(t.c_str(), &end, 10);
        if (end == t.c_str() || *end != '\0' || n < 1 || n > kMaxLanes)
            return fail(SECEDO_E_INVALID_ARG, "SECEDO_GPUS must be a count 1.." + std::to_string(kMaxLanes) + " or a comma-separated list of device ids");
        for (long i = 0; i < n; ++i) out->push_back(static_cast<int>(i));
        return SECEDO_OK;
    }
    size_t pos = 0;
    while (pos <= t.size()) {
        const size_t comma = std::min(t.find(',', pos), t.size());
        const std::string item = t.substr(pos, comma - pos);
        char *end = nullptr;
        const long v = std::strtol(item.c_str(), &end, 10);
        if (item.empty() || *end != '\0' || v < 0) return fail(SECEDO_E_INVALID_ARG, "SECEDO_GPUS: bad device id '" + item + "'");
        out->push_back(static_cast<int>(v));
        pos = comma + 1;
    }
    if (out->empty() || out->size() > static_cast<size_t>(kMaxLanes)) return fail(SECEDO_E_INVALID_ARG, "SECEDO_GPUS: 1.." + std::to_string(kMaxLanes) + " devices");
    return SECEDO_OK;
}

int check_devices(const std::vector<int> &ids) {
    const int n = secedo_simmat_device_count();
    if (n <= 0) return fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the similarity-matrix path has no CPU fallback");
    for (int d : ids)
        if (d < 0 || d >= n) return fail(SECEDO_E_NO_DEVICE, "device id " + std::to_string(d) + " out of range (" + std::to_string(n) + " visible)");
    return SECEDO_OK;
}

// the devices of the one-shot call
int one_shot_devices(std::vector<int> *out) {
    {
        std::lock_guard<std::mutex> lock(g_devices_mutex);
        *out = g_devices;
    }
    if (out->empty()) {
        if (const char *env = std::getenv("SECEDO_GPUS")) {
            const int rc = parse_device_list(env, out);
            if (rc != SECEDO_OK) return rc;
        } else {
            int device = 0;
            if (const char *env = std::getenv("SECEDO_DEVICE")) device = std::atoi(env);
            out->assign(1, device);
        }
    }
    if (out->size() > 1) return check_devices(*out);
    return SECEDO_OK;  // (one device: secedo_simmat_create checks it)
}

class HostBarrier {
public:
    explicit HostBarrier(unsigned n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lock(m_);
        const unsigned gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            ++gen_;
            cv_.notify_all();
        } else {
            cv_.wait(lock, [&] { return gen_ != gen; });
        }
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    unsigned n_, count_ = 0, gen_ = 0;
};

struct Lane {
    int device = 0;
    secedo_simmat_t *h = nullptr;
    hipStream_t s = nullptr, sc = nullptr;   // accumulation / exchange + normalisation
    std::vector<hipEvent_t> done;            // behind each chunk of the lane's tiles
    uint32_t lo = 0, hi = 0;                 // its tiles
    int rc = SECEDO_OK;
    std::string err;
};

constexpr uint32_t kChunks = 4;

int compute_on_devices(const std::vector<int> &devices, const uint32_t *chr_locus_off, uint32_t n_chr,
                       const uint32_t *locus_pos, const uint64_t *locus_entry_off, const uint32_t *read_ids,
                       const uint16_t *id_base16, const uint32_t *id_base32, const uint32_t *group_id_to_pos,
                       uint32_t n_groups, uint32_t num_cells, uint32_t mfl, double eps, double hr, double theta,
                       uint32_t num_threads, int normalization, double *out) {
    std::lock_guard<std::mutex> serial(g_multi_mutex);
    const uint32_t n = static_cast<uint32_t>(devices.size());
    std::vector<Lane> lanes(n);
    HostBarrier barrier(n);
    std::atomic<bool> failed{false};
    static const bool trace = std::getenv("SECEDO_ONE_SHOT_TRACE") != nullptr;
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    std::vector<double> t_ms(static_cast<size_t>(n) * 4, 0.0);

    auto lane_main = [&](uint32_t k) {
        Lane &me = lanes[k];
        me.device = devices[k];
        auto fail_lane = [&](int rc) {  // (the message is this thread's: keep it for the caller's thread)
            me.rc = rc;
            me.err = g_last_error;
            failed.store(true);
        };
#define LANE_HIP(expr)                                                         \
        do {                                                                   \
            hipError_t e__ = (expr);                                           \
            if (e__ != hipSuccess && me.rc == SECEDO_OK) fail_lane(hip_fail(e__, #expr)); \
        } while (0)
#define LANE_RC(expr)                                                          \
        do {                                                                   \
            if (me.rc == SECEDO_OK) {                                          \
                const int rc__ = (expr);                                       \
                if (rc__ != SECEDO_OK) fail_lane(rc__);                        \
            }                                                                  \
        } while (0)
        auto now_ms = [&] { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
        // ---- 1 + 2: pack, accumulate the lane's tiles chunk by chunk
        LANE_HIP(hipSetDevice(me.device));
        for (uint32_t j = 0; j < n && me.rc == SECEDO_OK; ++j) {
            if (devices[j] == me.device) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);  // direct xGMI copies where the link exists
            if (e != hipSuccess) (void)hipGetLastError();                   // (already enabled / not possible: staged copies)
        }
        LANE_RC(one_shot_acquire(kLaneKey + static_cast<int>(k), me.device, &me.h));
        LANE_HIP(hipStreamCreateWithFlags(&me.s, hipStreamNonBlocking));
        LANE_HIP(hipStreamCreateWithFlags(&me.sc, hipStreamNonBlocking));
        me.done.assign(kChunks, nullptr);
        for (uint32_t c = 0; c < kChunks; ++c) LANE_HIP(hipEventCreateWithFlags(&me.done[c], hipEventDisableTiming));
        LANE_RC(secedo_simmat_set_pileup(me.h, chr_locus_off, n_chr, locus_pos, locus_entry_off, read_ids, id_base16,
                                         id_base32, group_id_to_pos, n_groups));
        LANE_RC(secedo_simmat_prepare(me.h, num_cells, mfl, num_threads, 0, me.s));
        t_ms[k * 4 + 0] = now_ms();
        uint32_t tiles = 0, per = 0, step = 0;
        if (me.rc == SECEDO_OK) {
            tiles = me.h->num_tiles;
            per = (tiles + n - 1) / n;
            me.lo = std::min(k * per, tiles);
            me.hi = std::min((k + 1) * per, tiles);
            step = std::max(1u, (per + kChunks - 1) / kChunks);
            LANE_HIP(me.h->own_acc.ensure(std::max<uint64_t>(secedo_simmat_acc_elems(me.h), 1) * sizeof(int64_t)));
        }
        for (uint32_t c = 0; c < kChunks; ++c) {
            const uint32_t a = std::min(me.lo + c * step, me.hi), b = std::min(me.lo + (c + 1) * step, me.hi);
            // (an empty chunk still goes through accumulate once: it sets up the table and the scale finalize needs)
            if (b > a || c == 0) LANE_RC(secedo_simmat_assign(me.h, eps, hr, theta, a, b, me.h->own_acc.as<int64_t>(), me.s));
            if (me.rc == SECEDO_OK) LANE_HIP(hipEventRecord(me.done[c], me.s));
        }
        barrier.wait();  // every lane's events are recorded (a wait on an event not yet recorded would not wait)
        // ---- 3: the other lanes' tiles, chunk by chunk as their events fire
        if (!failed.load()) {
            const size_t b2 = static_cast<size_t>(me.h->pk.block_cells) * me.h->pk.block_cells;
            for (uint32_t d = 1; d < n; ++d) {  // (lane k starts with lane k + 1: the pulls spread over the links)
                const Lane &src = lanes[(k + d) % n];
                if (src.h->num_tiles != tiles || src.h->pk.block_cells != me.h->pk.block_cells) {
                    fail_lane(fail(SECEDO_E_STATE, "the lanes packed the same pileup into different geometries"));
                    break;
                }
                for (uint32_t c = 0; c < kChunks; ++c) {
                    const uint32_t a = std::min(src.lo + c * step, src.hi), b = std::min(src.lo + (c + 1) * step, src.hi);
                    if (b <= a) continue;
                    LANE_HIP(hipStreamWaitEvent(me.sc, src.done[c], 0));
                    LANE_HIP(hipMemcpyPeerAsync(me.h->own_acc.as<int64_t>() + a * b2, me.device,
                                                src.h->own_acc.as<int64_t>() + a * b2, src.device, (b - a) * b2 * sizeof(int64_t),
                                                me.sc));
                }
            }
            LANE_HIP(hipStreamWaitEvent(me.sc, me.done[kChunks - 1], 0));  // its own tiles
        }
        t_ms[k * 4 + 1] = now_ms();
        // ---- 4: this lane's rows, normalised here, downloaded from here
        if (!failed.load() && me.rc == SECEDO_OK) {
            const uint32_t row_lo = static_cast<uint32_t>(static_cast<uint64_t>(num_cells) * k / n);
            const uint32_t row_hi = static_cast<uint32_t>(static_cast<uint64_t>(num_cells) * (k + 1) / n);
            const size_t row_bytes = static_cast<size_t>(row_hi - row_lo) * num_cells * sizeof(double);
            if (me.h->scale_log2 != lanes[0].h->scale_log2)
                fail_lane(fail(SECEDO_E_STATE, "the lanes quantised the same pileup at different scales"));
            LANE_HIP(me.h->own_out.ensure(std::max<size_t>(row_bytes, 16)));
            LANE_RC(secedo_simmat_finalize_rows(me.h, normalization, me.h->own_acc.as<int64_t>(), row_lo, row_hi,
                                                me.h->own_out.as<double>(), me.sc));
            LANE_HIP(hipStreamSynchronize(me.sc));
            t_ms[k * 4 + 2] = now_ms();
            if (me.rc == SECEDO_OK && row_bytes) {
                double *dst = out + static_cast<size_t>(row_lo) * num_cells;
                if (row_bytes >= (8u << 20))
                    // (copier threads: the lanes share what one lane alone would use -- at least 16 in all: touching the
                    // fresh pages of the caller's matrix is what bounds a download, 2 GB/s for one thread)
                    LANE_HIP(download_pipelined(me.h->own_out.p, dst, row_bytes, std::max(1u, std::max(num_threads, 16u) / n),
                                                1 + static_cast<int>(k)));
                else
                    LANE_HIP(hipMemcpy(dst, me.h->own_out.p, row_bytes, hipMemcpyDeviceToHost));
            }
        } else if (me.sc) {
            (void)hipStreamSynchronize(me.sc);
        }
        t_ms[k * 4 + 3] = now_ms();
        barrier.wait();  // nobody reads this lane's accumulator any more
        if (me.s) (void)hipStreamSynchronize(me.s);
        if (me.h) me.h->uploads.settle();  // (their events were recorded on streams that end here)
        for (hipEvent_t ev : me.done)
            if (ev) (void)hipEventDestroy(ev);
        if (me.s) (void)hipStreamDestroy(me.s);
        if (me.sc) (void)hipStreamDestroy(me.sc);
        one_shot_release(kLaneKey + static_cast<int>(k), me.h, me.rc == SECEDO_OK && !failed.load());
        me.h = nullptr;
#undef LANE_HIP
#undef LANE_RC
    };

    std::vector<std::thread> threads;
    for (uint32_t k = 1; k < n; ++k) threads.emplace_back(lane_main, k);
    lane_main(0);
    for (std::thread &t : threads) t.join();
    if (trace)
        for (uint32_t k = 0; k < n; ++k)
            std::fprintf(stderr, "[one-shot, lane %u on device %d] packed at %.2f ms, exchange issued at %.2f, rows normalised at "
                                 "%.2f, downloaded at %.2f\n", k, lanes[k].device, t_ms[k * 4], t_ms[k * 4 + 1], t_ms[k * 4 + 2], t_ms[k * 4 + 3]);
    for (const Lane &lane : lanes)
        if (lane.rc != SECEDO_OK) return fail(lane.rc, lane.err);
    return SECEDO_OK;
}

}  // namespace

int secedo_simmat_set_devices(const int *device_ids, uint32_t n_devices) {
    if (n_devices && !device_ids) return fail(SECEDO_E_INVALID_ARG, "device_ids is null");
    if (n_devices > static_cast<uint32_t>(kMaxLanes)) return fail(SECEDO_E_LIMIT, "at most " + std::to_string(kMaxLanes) + " devices");
    std::vector<int> ids(device_ids, device_ids + n_devices);
    if (!ids.empty()) {
        const int rc = check_devices(ids);
        if (rc != SECEDO_OK) return rc;
    }
    std::lock_guard<std::mutex> lock(g_devices_mutex);
    g_devices = std::move(ids);
    return SECEDO_OK;
}

int secedo_simmat_get_devices(int *device_ids, uint32_t capacity) {
    std::vector<int> ids;
    const int rc = one_shot_devices(&ids);
    if (rc != SECEDO_OK) return rc;
    for (uint32_t i = 0; i < capacity && i < ids.size() && device_ids; ++i) device_ids[i] = ids[i];
    return static_cast<int>(ids.size());
}

int secedo_simmat_compute(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
                          const uint64_t *locus_entry_off, const uint32_t *read_ids,
                          const uint16_t *id_base16, const uint32_t *id_base32,
                          const uint32_t *group_id_to_pos, uint32_t n_groups, uint32_t num_cells,
                          uint32_t max_fragment_length, double mutation_rate, double homozygous_rate,
                          double seq_error_rate, uint32_t num_threads, int normalization, double *out) {
    if (normalization < 0 || normalization > 2)
        return fail(SECEDO_E_INVALID_NORMALIZATION, "Invalid normalization: " + std::to_string(normalization));
    if (!out) return fail(SECEDO_E_INVALID_ARG, "out is null");
    std::vector<int> devices;
    int rc = one_shot_devices(&devices);
    if (rc != SECEDO_OK) return rc;
    if (devices.size() > 1)
        return compute_on_devices(devices, chr_locus_off, n_chr, locus_pos, locus_entry_off, read_ids, id_base16, id_base32,
                                  group_id_to_pos, n_groups, num_cells, max_fragment_length, mutation_rate, homozygous_rate,
                                  seq_error_rate, num_threads, normalization, out);
    const int device = devices[0];
    // The caller of the reference's signature calls this once per sub-cluster of the recursion
    // (spectral_clustering.cpp:354-356): the handle with its device arenas, streams and tables is kept
    // between calls (two thirds of a first call on C2 is allocation). A failed call drops its handle.
    secedo_simmat_t *h = nullptr;
    rc = one_shot_acquire(device, device, &h);
    if (rc != SECEDO_OK) return rc;
    struct Guard {
        secedo_simmat_t *h;
        int device;
        bool ok = false;
        ~Guard() { one_shot_release(device, h, ok); }
    } guard{h, device};
    static const bool trace = std::getenv("SECEDO_ONE_SHOT_TRACE") != nullptr;  // phase times on stderr
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clock::now();
    rc = secedo_simmat_set_pileup(h, chr_locus_off, n_chr, locus_pos, locus_entry_off, read_ids,
                                  id_base16, id_base32, group_id_to_pos, n_groups);
    if (rc != SECEDO_OK) return rc;
    rc = secedo_simmat_prepare(h, num_cells, max_fragment_length, num_threads, 0, nullptr);
    if (rc != SECEDO_OK) return rc;
    const auto t1 = clock::now();
    HIP_TRY(h->own_acc.ensure(secedo_simmat_acc_elems(h) * sizeof(int64_t)));
    const size_t out_bytes = static_cast<size_t>(num_cells) * num_cells * sizeof(double);
    HIP_TRY(h->own_out.ensure(out_bytes));
    rc = secedo_simmat_assign_finalize(h, mutation_rate, homozygous_rate, seq_error_rate, normalization,
                                       h->own_acc.as<int64_t>(), h->own_out.as<double>(), nullptr);
    if (rc != SECEDO_OK) return rc;
    if (trace) HIP_TRY(hipDeviceSynchronize());
    const auto t2 = clock::now();
    // (a plain hipMemcpy into the caller's pageable, usually untouched matrix runs at 12-20 GB/s: 25-45 ms for
    // the 512 MB of C3, the longest phase of a one-shot call)
    static const bool plain_copy = [] { const char *e = std::getenv("SECEDO_ONE_SHOT_COPY"); return e && !std::strcmp(e, "plain"); }();
    if (out_bytes >= (8u << 20) && !plain_copy) {
        HIP_TRY(hipStreamSynchronize(nullptr));  // the matrix is complete (the download runs on a stream of its own)
        HIP_TRY(download_pipelined(h->own_out.p, out, out_bytes, std::max(1u, num_threads)));
    } else {
        HIP_TRY(hipMemcpy(out, h->own_out.p, out_bytes, hipMemcpyDeviceToHost));
    }
    if (trace)
        std::fprintf(stderr, "[one-shot] upload + packing %.2f ms, matrix %.2f ms, copy to the host %.2f ms\n", ms(t0, t1),
                     ms(t1, t2), ms(t2, clock::now()));
    guard.ok = true;
    return SECEDO_OK;
}

int secedo_simmat_staging_acquire(const uint64_t bytes[5], void *ptrs[5]) {
    if (!bytes || !ptrs) return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (secedo_simmat_device_count() <= 0) return fail(SECEDO_E_NO_DEVICE, "no HIP device is visible");
    std::lock_guard<std::mutex> lock(g_staging_mutex);
    if (g_staging_busy) return fail(SECEDO_E_STATE, "the staging buffers are held by another caller");
    for (int i = 0; i < 5; ++i) {
        const hipError_t e = g_staging[i].ensure(static_cast<size_t>(bytes[i]));
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc (staging)");
        ptrs[i] = g_staging[i].p;
    }
    g_staging_busy = true;
    return SECEDO_OK;
}

void secedo_simmat_staging_release(void) {
    std::lock_guard<std::mutex> lock(g_staging_mutex);
    g_staging_busy = false;
}

void secedo_simmat_release_cache(void) {
    secedo::em_release_cache();
    {
        std::lock_guard<std::mutex> lock(g_staging_mutex);
        if (!g_staging_busy)
            for (PinnedBuf &b : g_staging) b.release();
    }
    for (Bounce &b : g_bounce) {
        std::lock_guard<std::mutex> lock(b.mutex);
        b.buf.release();
    }
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &slot : g_pool) {
        if (slot.second) secedo_simmat_destroy(slot.second);
        slot.second = nullptr;
    }
}

// ------------------------------------------------------------------------------------------------
// Locus filter (reference util/is_significant.cpp)
// ------------------------------------------------------------------------------------------------

int secedo_is_significant(const uint16_t *base_count, double seq_error_rate, uint32_t cell_proportion) {
    if (!base_count) return fail(SECEDO_E_INVALID_ARG, "base_count is null");
    if (cell_proportion > 4) return fail(SECEDO_E_INVALID_ARG, "cell_proportion must be in [0, 4]");
    return secedo::is_significant(base_count, seq_error_rate, cell_proportion);
}

int secedo_filter_device(const uint32_t *d_chr_locus_off, uint32_t n_chr, const uint32_t *d_locus_pos,
                         const uint64_t *d_locus_entry_off, const uint32_t *d_read_ids,
                         const uint16_t *d_id_base16, const uint32_t *d_id_base32, const uint32_t *d_id_to_pos,
                         uint32_t n_groups, uint32_t n_loci, uint64_t n_entries, double seq_error_rate,
                         uint32_t cell_proportion, uint32_t *d_out_chr_locus_off, uint32_t *d_out_locus_pos,
                         uint64_t *d_out_locus_entry_off, uint32_t *d_out_read_ids, void *d_out_id_base,
                         uint64_t *out_n_loci, uint64_t *out_n_entries, double *avg_coverage, void *stream) {
    if (!d_chr_locus_off || !d_locus_entry_off || !d_out_chr_locus_off || !d_out_locus_entry_off || !out_n_loci
        || !out_n_entries || !avg_coverage)
        return fail(SECEDO_E_INVALID_ARG, "null argument");
    if (secedo_simmat_device_count() <= 0)
        return fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the locus filter has no CPU fallback");
    secedo::DeviceFlatPileup in;
    in.chr_locus_off = d_chr_locus_off;
    in.n_chr = n_chr;
    in.locus_pos = d_locus_pos;
    in.locus_entry_off = d_locus_entry_off;
    in.read_ids = d_read_ids;
    in.id_base16 = d_id_base16;
    in.id_base32 = d_id_base32;
    in.group_id_to_pos = d_id_to_pos;
    in.n_groups = n_groups;
    in.n_loci = n_loci;
    in.n_entries = n_entries;
    secedo::FilterOut out{d_out_chr_locus_off, d_out_locus_pos, d_out_locus_entry_off, d_out_read_ids, d_out_id_base};
    // scratch kept between calls, one set per device (allocations belong to the device they were made on)
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    static thread_local std::map<int, secedo::FilterWorkspace> workspaces;
    secedo::FilterWorkspace &ws = workspaces[device];
    const std::string err = secedo::filter_device(in, seq_error_rate, cell_proportion,
                                                  static_cast<hipStream_t>(stream), &ws, out, out_n_loci,
                                                  out_n_entries, avg_coverage);
    if (!err.empty()) return fail(err.find("hip") == 0 ? SECEDO_E_HIP : SECEDO_E_INVALID_ARG, err);
    return SECEDO_OK;
}

int secedo_filter(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
                  const uint64_t *locus_entry_off, const uint32_t *read_ids, const uint16_t *id_base16,
                  const uint32_t *id_base32, const uint32_t *id_to_pos, uint32_t n_groups,
                  double seq_error_rate, uint32_t cell_proportion, uint32_t *out_chr_locus_off,
                  uint32_t *out_locus_pos, uint64_t *out_locus_entry_off, uint32_t *out_read_ids,
                  void *out_id_base, uint64_t *out_n_loci, uint64_t *out_n_entries, double *avg_coverage) {
    if (!chr_locus_off || !locus_entry_off) return fail(SECEDO_E_INVALID_ARG, "null offset arrays");
    if ((id_base16 != nullptr) == (id_base32 != nullptr))
        return fail(SECEDO_E_INVALID_ARG, "exactly one of id_base16 / id_base32 must be given");
    if (secedo_simmat_device_count() <= 0)
        return fail(SECEDO_E_NO_DEVICE, "no HIP device is visible: the locus filter has no CPU fallback");
    int device = 0;
    if (const char *env = std::getenv("SECEDO_DEVICE")) device = std::atoi(env);
    HIP_TRY(hipSetDevice(device));
    const uint32_t L = chr_locus_off[n_chr];
    const uint64_t E = locus_entry_off[L];
    const size_t idw = id_base16 ? 2 : 4;
    DevBuf d_chr, d_pos, d_off, d_rid, d_idb, d_i2p, o_chr, o_pos, o_off, o_rid, o_idb;
    HIP_TRY(buf_upload(d_chr, chr_locus_off, (size_t)n_chr + 1));
    HIP_TRY(buf_upload(d_pos, locus_pos, L));
    HIP_TRY(buf_upload(d_off, locus_entry_off, (size_t)L + 1));
    HIP_TRY(buf_upload(d_rid, read_ids, E));
    if (id_base16) HIP_TRY(buf_upload(d_idb, id_base16, E));
    else HIP_TRY(buf_upload(d_idb, id_base32, E));
    HIP_TRY(buf_upload(d_i2p, id_to_pos, n_groups));
    HIP_TRY(o_chr.ensure(((size_t)n_chr + 1) * 4));
    HIP_TRY(o_pos.ensure((size_t)L * 4));
    HIP_TRY(o_off.ensure(((size_t)L + 1) * 8));
    HIP_TRY(o_rid.ensure(E * 4));
    HIP_TRY(o_idb.ensure(E * idw));
    const int rc = secedo_filter_device(
            d_chr.as<uint32_t>(), n_chr, d_pos.as<uint32_t>(), d_off.as<uint64_t>(), d_rid.as<uint32_t>(),
            id_base16 ? d_idb.as<uint16_t>() : nullptr, id_base16 ? nullptr : d_idb.as<uint32_t>(),
            d_i2p.as<uint32_t>(), n_groups, L, E, seq_error_rate, cell_proportion, o_chr.as<uint32_t>(),
            o_pos.as<uint32_t>(), o_off.as<uint64_t>(), o_rid.as<uint32_t>(), o_idb.p, out_n_loci, out_n_entries,
            avg_coverage, nullptr);
    if (rc != SECEDO_OK) return rc;
    HIP_TRY(hipMemcpy(out_chr_locus_off, o_chr.p, ((size_t)n_chr + 1) * 4, hipMemcpyDeviceToHost));
    if (*out_n_loci) HIP_TRY(hipMemcpy(out_locus_pos, o_pos.p, *out_n_loci * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_locus_entry_off, o_off.p, (*out_n_loci + 1) * 8, hipMemcpyDeviceToHost));
    if (*out_n_entries) {
        HIP_TRY(hipMemcpy(out_read_ids, o_rid.p, *out_n_entries * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_id_base, o_idb.p, *out_n_entries * idw, hipMemcpyDeviceToHost));
    }
    return SECEDO_OK;
}

}  // extern "C"
