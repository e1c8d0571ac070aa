// synth.cpp -- SYNTH-v1 synthetic fragment pileups. Bench / test utility,
// host only, deterministic for a given spec on every platform (splitmix64, integer thresholds;
// no std::*_distribution).
//
// Per chromosome: loci at gaps 1 + rng % gap_max; at every locus each cell starts a new fragment
// with probability new_frag_prob; a fragment has a length uniform in [frag_min, frag_max] base pairs
// and an entry at every locus inside [start, start + length). Bases: a per-locus reference base;
// cells of the second clone (cell >= num_cells / 2) carry a different base at every third locus;
// each base is replaced by a uniform one with probability base_error. A fraction mate_frac of the
// fragments gets a second (mate) entry at one of their first three loci, half of them with a
// conflicting base (exercises reference: similarity_matrix.cpp:387-395). Read ids are consecutive
// and restart at 0 in every chromosome (a read is identified by its id within a chromosome, :407).
#include "synth.h"

#include <cstring>
#include <vector>

namespace {

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};

uint64_t threshold(double p) {  // P(next() < threshold) = p
    if (p <= 0) return 0;
    if (p >= 1) return ~0ull;
    return static_cast<uint64_t>(p * 18446744073709551616.0);
}

struct Fragment {
    uint32_t id, cell, end;  // end: first position NOT covered
    uint32_t seen;           // loci covered so far
    uint32_t mate_at;        // locus ordinal that gets the mate entry, 0xFFFFFFFF = none
    bool mate_conflict;
};

struct Result {
    std::vector<uint32_t> chr_off, pos, rid, idb;
    std::vector<uint64_t> off;
};

thread_local Result g_result;
thread_local bool g_have = false;

void generate(const secedo_synth_spec &sp, Result &r) {
    r = Result();
    SplitMix64 rng{sp.seed};
    const uint64_t t_new = threshold(sp.new_frag_prob);
    const uint64_t t_err = threshold(sp.base_error);
    const uint64_t t_mate = threshold(sp.mate_frac);
    const uint32_t n_chr = sp.num_chromosomes ? sp.num_chromosomes : 1;
    const uint32_t span = sp.frag_max - sp.frag_min + 1;
    r.chr_off.push_back(0);
    r.off.push_back(0);
    std::vector<Fragment> live, keep;
    uint32_t done = 0;
    for (uint32_t c = 0; c < n_chr; ++c) {
        // loci are dealt to chromosomes as evenly as possible
        const uint32_t n_here = sp.num_loci / n_chr + (c < sp.num_loci % n_chr ? 1 : 0);
        uint32_t position = 1000;
        uint32_t next_id = 0;
        live.clear();
        for (uint32_t k = 0; k < n_here; ++k, ++done) {
            position += 1 + static_cast<uint32_t>(rng.next() % sp.gap_max);
            const uint32_t ref_base = static_cast<uint32_t>(rng.next() & 3u);
            const bool clone_locus = (done % 3u) == 0u;
            keep.clear();
            for (const Fragment &f : live) {
                if (f.end > position) keep.push_back(f);
            }
            live.swap(keep);
            for (uint32_t cell = 0; cell < sp.num_cells; ++cell) {
                if (rng.next() < t_new) {
                    Fragment f;
                    f.id = next_id++;
                    f.cell = cell;
                    f.end = position + sp.frag_min + static_cast<uint32_t>(rng.next() % span);
                    f.seen = 0;
                    f.mate_at = 0xFFFFFFFFu;
                    f.mate_conflict = false;
                    if (rng.next() < t_mate) {
                        f.mate_at = static_cast<uint32_t>(rng.next() % 3u);
                        f.mate_conflict = (rng.next() & 1u) != 0u;
                    }
                    live.push_back(f);
                }
            }
            for (Fragment &f : live) {
                uint32_t base = ref_base;
                if (clone_locus && f.cell >= sp.num_cells / 2) base = (base + 1u) & 3u;
                if (rng.next() < t_err) base = static_cast<uint32_t>(rng.next() & 3u);
                r.rid.push_back(f.id);
                r.idb.push_back((f.cell << 2) | base);
                if (f.seen == f.mate_at) {
                    const uint32_t b2 = f.mate_conflict
                            ? ((base + 1u + static_cast<uint32_t>(rng.next() % 3u)) & 3u)
                            : base;
                    r.rid.push_back(f.id);
                    r.idb.push_back((f.cell << 2) | b2);
                }
                ++f.seen;
            }
            r.pos.push_back(position);
            r.off.push_back(r.rid.size());
        }
        r.chr_off.push_back(static_cast<uint32_t>(r.pos.size()));
    }
}

}  // namespace

extern "C" int secedo_synth_generate(const secedo_synth_spec *spec, uint64_t *n_loci,
                                     uint64_t *n_entries, uint32_t *chr_locus_off,
                                     uint32_t *locus_pos, uint64_t *locus_entry_off,
                                     uint32_t *read_ids, uint32_t *id_base32) {
    if (!spec || !n_loci || !n_entries) return -1;
    if (spec->num_cells == 0 || spec->num_cells > 65535 || spec->gap_max == 0
        || spec->frag_max < spec->frag_min || spec->frag_min == 0)
        return -1;
    const bool sizing = !chr_locus_off && !locus_pos && !locus_entry_off && !read_ids && !id_base32;
    if (sizing || !g_have) {
        generate(*spec, g_result);
        g_have = true;
    }
    *n_loci = g_result.pos.size();
    *n_entries = g_result.rid.size();
    if (sizing) return 0;
    if (!chr_locus_off || !locus_pos || !locus_entry_off || !read_ids || !id_base32)
        return -1;
    std::memcpy(chr_locus_off, g_result.chr_off.data(), g_result.chr_off.size() * sizeof(uint32_t));
    std::memcpy(locus_pos, g_result.pos.data(), g_result.pos.size() * sizeof(uint32_t));
    std::memcpy(locus_entry_off, g_result.off.data(), g_result.off.size() * sizeof(uint64_t));
    std::memcpy(read_ids, g_result.rid.data(), g_result.rid.size() * sizeof(uint32_t));
    std::memcpy(id_base32, g_result.idb.data(), g_result.idb.size() * sizeof(uint32_t));
    g_result = Result();
    g_have = false;
    return 0;
}
