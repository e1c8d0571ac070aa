#include "pack_host.hpp"

#include <algorithm>
#include <cstring>

namespace secedo {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct Segment {          // one live read (reference: struct Read, similarity_matrix.cpp:172-182)
    uint32_t start;       // position of the first entry, kept even if that entry is removed
    uint32_t cell;        // group_id_to_pos[group]
    uint32_t last_kept;   // entry index of the newest kept entry, kNone if the read is empty
    bool tail;            // never flushed: still live when its chromosome ended
};

// open-addressing map read_id -> segment index, reset per chromosome
struct IdMap {
    std::vector<uint32_t> key, val;
    std::vector<uint8_t> used;
    uint32_t mask = 0;
    void reset(uint64_t expected) {
        uint64_t want = 16;
        while (want < 2 * expected + 2) want <<= 1;
        if (want > key.size()) {
            key.resize(want);
            val.resize(want);
            used.resize(want);
        }
        mask = static_cast<uint32_t>(key.size() - 1);
        std::fill(used.begin(), used.end(), 0);
    }
    static uint32_t mix(uint32_t x) {
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        return x;
    }
    uint32_t slot(uint32_t id) const {
        uint32_t s = mix(id) & mask;
        while (used[s] && key[s] != id) s = (s + 1) & mask;
        return s;
    }
};

}  // namespace

std::string pack_pileup(const FlatPileupView &in, uint32_t num_cells, uint32_t mfl,
                        uint32_t num_threads, uint32_t block_cells,
                        StageGeometry (*geometry)(uint32_t block_cells), bool allow_count_tile,
                        PackedPileup *out) {
    if (!in.chr_locus_off || !in.locus_entry_off) return "null pileup arrays";
    if ((in.id_base16 != nullptr) == (in.id_base32 != nullptr))
        return "exactly one of id_base16 / id_base32 must be given";
    if (num_threads == 0) return "num_threads must be positive";
    if (num_cells == 0 || num_cells > 65535) return "num_cells must be in [1, 65535]";
    if (block_cells != 0 && block_cells != 64 && block_cells != 128) return "block_cells must be 0, 64 or 128";
    const uint32_t L = in.n_loci();
    const uint64_t E = in.n_entries();
    if (E >= 0xFFFFFFF0ull) return "more than 2^32 pileup entries are not supported";
    if (L && (!in.locus_pos || !in.read_ids)) return "null pileup arrays";

    PackedPileup &pk = *out;
    pk = PackedPileup();
    pk.num_cells = num_cells;
    pk.num_loci = L;
    pk.raw_entries = E;

    // ---- pass 1: read assembly + flush schedule, entry by entry (input order) ---------------
    std::vector<uint32_t> seg_of(E), prev_kept(E, kNone), locus_of(E);
    std::vector<uint8_t> kept(E, 0);
    std::vector<Segment> segs;
    segs.reserve(E / 2 + 16);
    IdMap map;
    uint32_t completed = 0;  // NOT reset per chromosome (similarity_matrix.cpp:344)
    const uint64_t flush_at = 4ull * num_threads;  // :354-356
    for (uint32_t c = 0; c < in.n_chr; ++c) {
        const uint32_t l0 = in.chr_locus_off[c], l1 = in.chr_locus_off[c + 1];
        if (l1 < l0 || l1 > L) return "chr_locus_off is not monotone";
        map.reset(l1 > l0 ? in.locus_entry_off[l1] - in.locus_entry_off[l0] : 0);
        uint32_t front = static_cast<uint32_t>(segs.size());  // first live segment
        for (uint32_t l = l0; l < l1; ++l) {
            const uint32_t position = in.locus_pos[l];
            // the reference merge-walks reads by position (similarity_matrix.cpp:223-229) and
            // asserts non-decreasing positions (:398): loci must be strictly increasing
            if (l > l0 && position <= in.locus_pos[l - 1])
                return "positions must be strictly increasing within a chromosome";
            const uint32_t live = static_cast<uint32_t>(segs.size()) - front;
            // :348-352, uint32 arithmetic as in the reference
            for (uint32_t i = completed;
                 i < live && static_cast<uint32_t>(segs[front + i].start + mfl) <= position; ++i) {
                ++completed;
            }
            if (completed >= flush_at) {  // :356-373: these reads are compared and erased
                front += completed;
                completed = 0;
            }
            const uint64_t e0 = in.locus_entry_off[l], e1 = in.locus_entry_off[l + 1];
            if (e1 < e0 || e1 > E) return "locus_entry_off is not monotone";
            for (uint64_t e = e0; e < e1; ++e) {
                const uint32_t packed = in.id_base(e);
                const uint32_t group = packed >> 2;
                if (group >= in.n_groups) return "group id outside group_id_to_pos";
                const uint32_t cell = in.group_id_to_pos[group];
                if (cell >= num_cells) return "group_id_to_pos maps outside the matrix";
                const uint32_t rid = in.read_ids[e];
                const uint32_t s = map.slot(rid);
                locus_of[e] = l;
                if (!map.used[s] || map.val[s] < front) {  // :379-382 new (or re-opened) read
                    map.used[s] = 1;
                    map.key[s] = rid;
                    map.val[s] = static_cast<uint32_t>(segs.size());
                    segs.push_back(Segment{position, cell, static_cast<uint32_t>(e), false});
                    seg_of[e] = map.val[s];
                    kept[e] = 1;
                    continue;
                }
                Segment &sg = segs[map.val[s]];
                seg_of[e] = map.val[s];
                if (sg.last_kept != kNone && locus_of[sg.last_kept] == l) {
                    // :387-395 second mate at the same position: equal base -> ignore;
                    // different base -> drop the stored one as well
                    const uint32_t stored = in.id_base(sg.last_kept) & 3u;
                    if (stored != (packed & 3u)) {
                        kept[sg.last_kept] = 0;
                        sg.last_kept = prev_kept[sg.last_kept];
                    }
                    continue;
                }
                prev_kept[e] = sg.last_kept;  // :400-401 append
                sg.last_kept = static_cast<uint32_t>(e);
                kept[e] = 1;
            }
        }
        // :407-408 whatever is still live now is dropped without having been the earlier read
        for (uint32_t i = front; i < segs.size(); ++i) segs[i].tail = true;
    }

    // ---- pass 2: per live read, its kept entries in locus order + the window masks ----------
    const uint32_t R = static_cast<uint32_t>(segs.size());
    pk.num_reads = R;
    pk.read_off.assign(static_cast<size_t>(R) + 1, 0);
    uint64_t n_kept = 0;
    for (uint64_t e = 0; e < E; ++e) {
        if (kept[e]) {
            pk.read_off[seg_of[e] + 1]++;
            ++n_kept;
        }
    }
    for (uint32_t r = 0; r < R; ++r) pk.read_off[r + 1] += pk.read_off[r];
    pk.num_entries = n_kept;
    pk.read_locus.resize(n_kept);
    pk.read_base.resize(n_kept);
    // kept entries arrive in input order == locus order within a read
    std::vector<uint32_t> slot_of(E, kNone);  // entry -> index in read_locus
    {
        std::vector<uint32_t> fill(pk.read_off.begin(), pk.read_off.end() - 1);
        for (uint64_t e = 0; e < E; ++e) {
            if (!kept[e]) continue;
            const uint32_t k = fill[seg_of[e]]++;
            pk.read_locus[k] = locus_of[e];
            pk.read_base[k] = static_cast<uint8_t>(in.id_base(e) & 3u);
            slot_of[e] = k;
        }
    }

    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t n = pk.read_off[r + 1] - pk.read_off[r];
        if (n > 1) pk.multi_entries += n;
        pk.max_read_entries = std::max(pk.max_read_entries, n);
    }
    if (block_cells == 0) {
        const StageGeometry g64 = geometry(64);
        const bool clustered = n_kept && static_cast<double>(pk.multi_entries) > g64.masks_threshold * static_cast<double>(n_kept);
        block_cells = (clustered || num_cells <= 64) ? 64 : 128;
    }
    const StageGeometry geo = geometry(block_cells);
    pk.block_cells = block_cells;
    pk.num_blocks = (num_cells + block_cells - 1) / block_cells;

    // ---- pass 3: bin by (cell block, locus), emit the entry records --------------------------
    const uint32_t B = block_cells, nb = pk.num_blocks;
    const size_t stride = static_cast<size_t>(L) + 1;
    pk.blk_off.assign(static_cast<size_t>(nb) * stride, 0);
    for (uint64_t e = 0; e < E; ++e) {
        if (kept[e]) pk.blk_off[static_cast<size_t>(segs[seg_of[e]].cell / B) * stride + locus_of[e]]++;
    }
    {
        uint32_t run = 0;
        for (size_t i = 0; i < pk.blk_off.size(); ++i) {  // exclusive scan, block-major
            // the slot [b][L] holds no count: it becomes the end offset of block b
            const uint32_t c = pk.blk_off[i];
            pk.blk_off[i] = run;
            run += c;
        }
    }
    // ---- pair bound: max over cells of sum over loci of (entries of the cell at the locus)^2 -----
    {
        std::vector<uint64_t> sq(num_cells, 0);
        std::vector<uint32_t> cnt(num_cells, 0);
        std::vector<uint32_t> seen;
        for (uint32_t l = 0; l < L; ++l) {
            seen.clear();
            for (uint64_t e = in.locus_entry_off[l]; e < in.locus_entry_off[l + 1]; ++e) {
                if (!kept[e]) continue;
                const uint32_t cell = segs[seg_of[e]].cell;
                if (cnt[cell]++ == 0) seen.push_back(cell);
            }
            for (uint32_t cell : seen) {
                sq[cell] += static_cast<uint64_t>(cnt[cell]) * cnt[cell];
                cnt[cell] = 0;
            }
        }
        pk.pair_bound = 0;
        for (uint64_t v : sq) pk.pair_bound = std::max(pk.pair_bound, v);
        pk.cell_sq = std::move(sq);
    }

    // ---- pass 3b: locus ranges for LDS staging (greedy; one partition shared by all blocks) ----
    pk.stage_masks = n_kept && static_cast<double>(pk.multi_entries) > geo.masks_threshold * static_cast<double>(n_kept);
    pk.count_tile = allow_count_tile && !pk.stage_masks && pk.pair_bound < kCountTileLimit;
    pk.cap_entries = pk.stage_masks ? geo.cap_entries_masks : pk.count_tile ? geo.cap_entries_counts : geo.cap_entries_plain;
    pk.cap_loci = pk.stage_masks ? geo.cap_loci_masks : pk.count_tile ? geo.cap_loci_counts : geo.cap_loci_plain;
    if (pk.cap_entries == 0 || pk.cap_loci == 0 || pk.cap_loci > 65535 || pk.cap_entries > 65535)
        return "invalid staging geometry";
    pk.range_off.clear();
    pk.range_off.push_back(0);
    {
        std::vector<uint32_t> begin_off(nb);  // blk_off[b][range begin]
        auto restart = [&](uint32_t l) {
            for (uint32_t b = 0; b < nb; ++b) begin_off[b] = pk.blk_off[static_cast<size_t>(b) * stride + l];
        };
        auto fits_through = [&](uint32_t begin, uint32_t l) {  // range [begin, l + 1)
            if (l + 1 - begin > pk.cap_loci) return false;
            for (uint32_t b = 0; b < nb; ++b) {
                if (pk.blk_off[static_cast<size_t>(b) * stride + l + 1] - begin_off[b] > pk.cap_entries) return false;
            }
            return true;
        };
        uint32_t begin = 0;
        restart(0);
        for (uint32_t l = 0; l < L; ++l) {
            if (fits_through(begin, l)) continue;
            if (l > begin) {  // close [begin, l) and start over at l
                pk.range_off.push_back(l);
                begin = l;
                restart(l);
                if (fits_through(begin, l)) continue;
            }
            // locus l alone exceeds the cap: a single-locus range the kernel reads from HBM
            pk.range_off.push_back(l + 1);
            begin = l + 1;
            restart(l + 1);
        }
        if (pk.range_off.back() != L) pk.range_off.push_back(L);
    }
    std::vector<uint32_t> range_begin_of(L);  // first locus of the range a locus belongs to
    for (size_t r = 0; r + 1 < pk.range_off.size(); ++r) {
        for (uint32_t l = pk.range_off[r]; l < pk.range_off[r + 1]; ++l) range_begin_of[l] = pk.range_off[r];
    }

    pk.entry.resize(n_kept);
    pk.entry32.resize(n_kept);
    pk.n_wide = 0;
    pk.mask32.resize(n_kept);
    pk.entry_read.resize(n_kept);
    std::vector<uint32_t> cursor(pk.blk_off);  // next free index per (block, locus)
    for (uint32_t l = 0; l < L; ++l) {
        for (uint64_t e = in.locus_entry_off[l]; e < in.locus_entry_off[l + 1]; ++e) {
            if (!kept[e]) continue;
            const Segment &sg = segs[seg_of[e]];
            const uint32_t r = seg_of[e];
            const uint32_t k = slot_of[e];
            const uint32_t lo = pk.read_off[r], hi = pk.read_off[r + 1];
            const uint32_t base = pk.read_base[k];
            Entry a;
            a.meta = sg.cell | (base << kMetaBaseShift) | (sg.tail ? kMetaTail : 0u);
            a.masks = 0;
            a.bases = 0;
            a.locus = l;
            bool wide = false;
            for (uint32_t j = k; j-- > lo;) {
                const uint32_t dist = l - pk.read_locus[j];  // >= 1
                if (dist > kNarrowWindow) wide = true;
                if (dist > kWindow) {
                    a.meta |= kMetaPrevOvf;
                    break;
                }
                a.masks |= 1u << (dist - 1);
            }
            for (uint32_t j = k + 1; j < hi; ++j) {
                const uint32_t dist = pk.read_locus[j] - l;
                if (dist > kNarrowWindow) wide = true;
                if (dist > kWindow) {
                    a.meta |= kMetaNextOvf;
                    break;
                }
                a.masks |= 1u << (16 + dist - 1);
                a.bases |= static_cast<uint32_t>(pk.read_base[j] & 1u) << (dist - 1);
                a.bases |= static_cast<uint32_t>((pk.read_base[j] >> 1) & 1u) << (16 + dist - 1);
            }
            if (a.meta & (kMetaPrevOvf | kMetaNextOvf)) pk.any_window_overflow = true;
            const uint32_t blk = sg.cell / B;
            const uint32_t dst = cursor[static_cast<size_t>(blk) * stride + l]++;
            pk.entry[dst] = a;
            pk.entry_read[dst] = r;
            pk.n_wide += wide ? 1u : 0u;
            pk.entry32[dst] = (sg.cell - blk * B) | (base << kC_BaseShift) | (sg.tail ? kC_Tail : 0u)
                    | (hi - lo > 1 ? kC_Multi : 0u) | (wide ? kC_Wide : 0u)
                    | ((l - range_begin_of[l]) << 16);
            pk.mask32[dst] = (a.masks & 0xFFu) | (((a.masks >> 16) & 0xFFu) << 8)
                    | ((a.bases & 0xFFu) << 16) | (((a.bases >> 16) & 0xFFu) << 24);
        }
    }
    return std::string();
}

}  // namespace secedo
