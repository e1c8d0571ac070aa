// pileup_io.cpp -- the reference's pileup files straight into the flat structure-of-arrays layout
// of secedo_simmat.h, without the vector<PosData> intermediate.
//
// Formats (reference util/pileup_reader.cpp):
//   binary (*.bin, read_pileup_bin :139-257; writer pileup.cpp:328-333 and :107-116 of the text reader)
//       per locus: u32 position, u16 coverage, u32 read_ids[coverage], u16 cell_id << 2 | base [coverage]
//   text (read_pileup_text :12-137), tab separated:
//       chromosome, position, coverage, bases, comma separated cell ids, comma separated read ids
// Semantics kept: loci with more than max_coverage entries are skipped (:50, :190); an optional sorted
// list of positions restricts the loci (:54-66, :194-206); cell ids are mapped through id_to_group
// (:93, :219); text read ids are numbered in order of first appearance (:76-81); num_cells is the
// number of distinct cell ids (text, :136) resp. the largest cell id + 1 (binary, :232); the longest
// fragment is the largest (last - first) position of a read id (:121-129, :236-243), 1000 when not
// asked for in the binary reader (:256). Where the reference exits the process (missing file, cell id
// outside id_to_group) an error code is returned instead.
#include "secedo_simmat.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Parsed {
    std::string key;
    std::vector<uint32_t> pos, rid;
    std::vector<uint64_t> off;
    std::vector<uint16_t> idb;
    uint32_t num_cells = 0, max_len = 0;
};

thread_local Parsed g_cache;
thread_local bool g_have = false;
thread_local std::string g_io_error;

// base letter -> 0..3, anything else 5 (reference util/util.hpp:17-22 CharToInt; like the reference the
// 5 is OR-ed into the packed value unchecked)
inline uint16_t base_code(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;  // the table maps T/U and t/u to 3
        default: return 5;
    }
}

struct PositionFilter {
    const uint32_t *p;
    uint64_t n, idx = 0;
    // 1 keep, 0 skip, -1 stop (all listed positions seen)
    int check(uint32_t position) {
        if (n == 0) return 1;
        while (idx < n && p[idx] < position) ++idx;
        if (idx == n) return -1;
        return p[idx] > position ? 0 : 1;
    }
};

uint32_t longest_span(const std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> &stats) {
    uint32_t best = 0;
    for (const auto &kv : stats) best = std::max(best, kv.second.second - kv.second.first);
    return best;
}

int read_bin(const char *path, const uint16_t *id_to_group, uint32_t n_ids, uint32_t max_coverage,
             PositionFilter pf, bool want_len, Parsed *out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        g_io_error = std::string("File ") + path + " does not exist or is not readable.";
        return SECEDO_E_INVALID_ARG;
    }
    std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> stats;
    uint32_t max_cell = 0;
    std::vector<uint32_t> ids;
    std::vector<uint16_t> packed;
    out->off.push_back(0);
    while (true) {
        uint32_t position;
        uint16_t coverage;
        if (!f.read(reinterpret_cast<char *>(&position), 4)) break;
        if (!f.read(reinterpret_cast<char *>(&coverage), 2)) break;
        ids.resize(coverage);
        packed.resize(coverage);
        f.read(reinterpret_cast<char *>(ids.data()), coverage * 4);
        f.read(reinterpret_cast<char *>(packed.data()), coverage * 2);
        if (!f && coverage) {
            g_io_error = "truncated binary pileup record";
            return SECEDO_E_INVALID_ARG;
        }
        if (coverage > max_coverage) continue;
        const int keep = pf.check(position);
        if (keep < 0) break;
        if (keep == 0) continue;
        for (uint32_t i = 0; i < coverage; ++i) {
            const uint16_t cell = packed[i] >> 2;
            if (cell >= n_ids) {
                g_io_error = "Cell id " + std::to_string(cell) + " is too large for the id_to_group mapping";
                return SECEDO_E_INVALID_ARG;
            }
            max_cell = std::max<uint32_t>(max_cell, cell);
            packed[i] = static_cast<uint16_t>(id_to_group[cell] << 2 | (packed[i] & 3));
            if (want_len) {
                auto it = stats.find(ids[i]);
                if (it == stats.end()) stats[ids[i]] = {position, position};
                else it->second.second = position;
            }
        }
        out->pos.push_back(position);
        out->rid.insert(out->rid.end(), ids.begin(), ids.end());
        out->idb.insert(out->idb.end(), packed.begin(), packed.end());
        out->off.push_back(out->rid.size());
    }
    out->num_cells = max_cell + 1;
    out->max_len = want_len ? longest_span(stats) : 1000;
    return SECEDO_OK;
}

void split(const std::string &s, char sep, std::vector<std::string> *parts) {
    parts->clear();
    size_t b = 0;
    while (b <= s.size()) {
        const size_t e = s.find(sep, b);
        if (e == std::string::npos) {
            if (b < s.size()) parts->push_back(s.substr(b));  // std::getline drops an empty last field
            break;
        }
        parts->push_back(s.substr(b, e - b));
        b = e + 1;
    }
}

int read_text(const char *path, const uint16_t *id_to_group, uint32_t n_ids, uint32_t max_coverage,
              PositionFilter pf, bool write_bin, Parsed *out) {
    std::ifstream f(path);
    if (!f) {
        g_io_error = std::string("File ") + path + " does not exist or is not readable.";
        return SECEDO_E_INVALID_ARG;
    }
    std::ofstream bin;
    if (write_bin) bin.open(std::string(path) + ".bin", std::ios::binary);
    std::unordered_map<std::string, uint32_t> id_map;
    std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> stats;
    std::unordered_set<uint32_t> cells;
    std::vector<std::string> cols, parts;
    std::vector<uint32_t> ids;
    std::vector<uint16_t> cell_ids, grouped, raw;
    std::string line;
    out->off.push_back(0);
    while (std::getline(f, line)) {
        split(line, '\t', &cols);
        if (cols.size() < 6) {
            g_io_error = "pileup line with fewer than 6 tab separated columns";
            return SECEDO_E_INVALID_ARG;
        }
        const uint32_t position = static_cast<uint32_t>(std::strtoll(cols[1].c_str(), nullptr, 10));
        const std::string &bases = cols[3];
        split(cols[4], ',', &parts);
        cell_ids.clear();
        for (const std::string &p : parts) cell_ids.push_back(static_cast<uint16_t>(std::strtoll(p.c_str(), nullptr, 10)));
        if (cell_ids.size() > max_coverage) continue;
        const int keep = pf.check(position);
        if (keep < 0) break;
        if (keep == 0) continue;
        split(cols[5], ',', &parts);
        if (cell_ids.size() < bases.size() || parts.size() < bases.size()) {
            g_io_error = "pileup line with fewer cell or read ids than bases";
            return SECEDO_E_INVALID_ARG;
        }
        ids.resize(parts.size());
        for (size_t j = 0; j < parts.size(); ++j) {
            auto it = id_map.find(parts[j]);
            if (it == id_map.end()) it = id_map.emplace(parts[j], static_cast<uint32_t>(id_map.size())).first;
            ids[j] = it->second;
        }
        grouped.clear();
        raw.clear();
        for (size_t j = 0; j < bases.size(); ++j) {
            if (cell_ids[j] >= n_ids) {
                g_io_error = "Cell id " + std::to_string(cell_ids[j]) + " is too large for the id_to_group mapping";
                return SECEDO_E_INVALID_ARG;
            }
            const uint16_t code = base_code(static_cast<unsigned char>(bases[j]));
            grouped.push_back(static_cast<uint16_t>(id_to_group[cell_ids[j]] << 2 | code));
            raw.push_back(static_cast<uint16_t>(cell_ids[j] << 2 | code));
        }
        for (uint16_t c : cell_ids) cells.insert(c);
        for (size_t j = 0; j < ids.size(); ++j) {
            auto it = stats.find(ids[j]);
            if (it == stats.end()) stats[ids[j]] = {position, position};
            else it->second.second = position;
        }
        out->pos.push_back(position);
        out->rid.insert(out->rid.end(), ids.begin(), ids.end());
        out->idb.insert(out->idb.end(), grouped.begin(), grouped.end());
        // the reference keeps read_ids and cell_ids_and_bases of possibly different length in one
        // PosData; the flat layout needs one count per locus: the number of bases
        out->rid.resize(out->idb.size());
        out->off.push_back(out->idb.size());
        if (bin.is_open()) {
            const uint16_t coverage = static_cast<uint16_t>(ids.size());
            bin.write(reinterpret_cast<const char *>(&position), 4);
            bin.write(reinterpret_cast<const char *>(&coverage), 2);
            bin.write(reinterpret_cast<const char *>(ids.data()), ids.size() * 4);
            bin.write(reinterpret_cast<const char *>(raw.data()), raw.size() * 2);
        }
    }
    out->num_cells = static_cast<uint32_t>(cells.size());
    out->max_len = longest_span(stats);
    return SECEDO_OK;
}

bool ends_with(const std::string &s, const char *suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

}  // namespace

extern "C" const char *secedo_pileup_last_error(void) { return g_io_error.c_str(); }

extern "C" int secedo_pileup_read(const char *path, const uint16_t *id_to_group, uint32_t n_ids,
                                  uint32_t max_coverage, const uint32_t *positions, uint64_t n_positions,
                                  int compute_max_read_len, int write_bin, secedo_pileup_info *info,
                                  uint32_t *locus_pos, uint64_t *locus_entry_off, uint32_t *read_ids,
                                  uint16_t *id_base16) {
    if (!path || !info || (!id_to_group && n_ids)) {
        g_io_error = "null argument";
        return SECEDO_E_INVALID_ARG;
    }
    const bool sizing = !locus_pos && !locus_entry_off && !read_ids && !id_base16;
    const std::string key = std::string(path) + "|" + std::to_string(max_coverage) + "|" + std::to_string(n_positions)
            + "|" + std::to_string(compute_max_read_len) + "|" + std::to_string(n_ids);
    if (sizing || !g_have || g_cache.key != key) {
        g_cache = Parsed();
        g_cache.key = key;
        PositionFilter pf{positions, n_positions};
        const int rc = ends_with(path, ".bin")
                ? read_bin(path, id_to_group, n_ids, max_coverage, pf, compute_max_read_len != 0, &g_cache)
                : read_text(path, id_to_group, n_ids, max_coverage, pf, write_bin != 0, &g_cache);
        if (rc != SECEDO_OK) {
            g_have = false;
            return rc;
        }
        g_have = true;
    }
    info->n_loci = g_cache.pos.size();
    info->n_entries = g_cache.rid.size();
    info->num_cells = g_cache.num_cells;
    info->max_read_length = g_cache.max_len;
    if (sizing) return SECEDO_OK;
    if (!locus_pos || !locus_entry_off || !read_ids || !id_base16) {
        g_io_error = "all four output arrays are needed";
        return SECEDO_E_INVALID_ARG;
    }
    std::memcpy(locus_pos, g_cache.pos.data(), g_cache.pos.size() * 4);
    std::memcpy(locus_entry_off, g_cache.off.data(), g_cache.off.size() * 8);
    std::memcpy(read_ids, g_cache.rid.data(), g_cache.rid.size() * 4);
    std::memcpy(id_base16, g_cache.idb.data(), g_cache.idb.size() * 2);
    g_cache = Parsed();
    g_have = false;
    return SECEDO_OK;
}
