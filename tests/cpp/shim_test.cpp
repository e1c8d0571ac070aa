// shim_test.cpp -- pure C++ host (no Python, no torch) through include/secedo_simmat.hpp.
//
// Stand-ins for the two reference types the shim is templated on, written for this test:
// `PosData` carries the three members of the reference's per-locus record
// (reference: sequenced_data.hpp:26-37); `Mat` the owning row-major matrix subset the boundary
// returns (reference: util/mat.hpp:86 constructor, :117 element access, :238 data(), rows()/cols());
// `MatNoData` the same without data(), which sends the shim through operator() instead.
// (tests/test_reference_headers_cpu.py compiles the same shim against the reference's real headers.)
//
// usage: shim_test <pileup.bin> <num_cells> <mfl> <num_threads> <normalization> <out.f64> [consumers|nodata]
//   pileup.bin = the reference's binary pileup records (u32 position, u16 coverage,
//   u32 read_ids[coverage], u16 id_base[coverage]; util/pileup_reader.cpp:166-179), one chromosome.
//   "consumers": the 7 smallest eigenpairs of the matrix and the EM refinement started from the sign of
//   the second eigenvector (secedo_pipeline.hpp); appended to out.f64 are 7 eigenvalues, 7 eigenvectors
//   (column-major) and the refined probabilities.
//
// usage: shim_test --synth <cells> <loci> <chromosomes> <gap_max> <new_frag_prob> [repeats]
//   times the drop-in call on a SYNTH-v1 pileup held as vector<vector<PosData>> (what the reference's
//   caller holds): prints one JSON line with the wall-clock of the first and of the repeated call
//   (flatten + H2D + device step + D2H into the returned matrix) and a checksum. bench.py runs it for
//   the headline workload (`cpp_dropin_call`).
#include "secedo_pipeline.hpp"
#include "secedo_simmat.hpp"
#include "synth.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <vector>

struct PosData {
    uint32_t position;
    std::vector<uint32_t> read_ids;
    std::vector<uint16_t> group_ids_bases;
};

class Mat {
  public:
    Mat(uint32_t r, uint32_t c) : r_(r), c_(c), el_(new double[static_cast<size_t>(r) * c]) {}
    Mat(Mat &&) = default;
    Mat(const Mat &) = delete;
    double &operator()(uint32_t i, uint32_t j) { return el_[static_cast<size_t>(i) * c_ + j]; }
    uint32_t rows() const { return r_; }
    uint32_t cols() const { return c_; }
    double *data() { return el_.get(); }
    const double *data() const { return el_.get(); }

  private:
    uint32_t r_, c_;
    std::unique_ptr<double[]> el_;
};

class MatNoData {
  public:
    MatNoData(uint32_t r, uint32_t c) : r_(r), c_(c), el_(static_cast<size_t>(r) * c) {}
    double &operator()(uint32_t i, uint32_t j) { return el_[static_cast<size_t>(i) * c_ + j]; }
    const std::vector<double> &elements() const { return el_; }

  private:
    uint32_t r_, c_;
    std::vector<double> el_;
};

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int run_synth(int argc, char **argv) {
    secedo_synth_spec spec;
    std::memset(&spec, 0, sizeof(spec));
    spec.num_cells = static_cast<uint32_t>(std::atoi(argv[2]));
    spec.num_loci = static_cast<uint32_t>(std::atoi(argv[3]));
    spec.num_chromosomes = static_cast<uint32_t>(std::atoi(argv[4]));
    spec.gap_max = static_cast<uint32_t>(std::atoi(argv[5]));
    spec.new_frag_prob = std::atof(argv[6]);
    spec.frag_min = 50;
    spec.frag_max = 600;
    spec.base_error = 0.01;
    spec.mate_frac = 0.01;
    spec.seed = 42;
    const int repeats = argc > 7 ? std::atoi(argv[7]) : 2;
    if (spec.num_cells > 16383) {
        std::fprintf(stderr, "the reference's PosData holds 14-bit group ids (sequenced_data.hpp:29-37)\n");
        return 2;
    }
    uint64_t L = 0, E = 0;
    if (secedo_synth_generate(&spec, &L, &E, nullptr, nullptr, nullptr, nullptr, nullptr) != 0) return 2;
    std::vector<uint32_t> chr(spec.num_chromosomes + 1), pos(L), rid(E), idb(E);
    std::vector<uint64_t> off(L + 1);
    if (secedo_synth_generate(&spec, &L, &E, chr.data(), pos.data(), off.data(), rid.data(), idb.data()) != 0) return 2;
    // the caller's representation: one vector<PosData> per chromosome (three heap blocks per locus)
    std::vector<std::vector<PosData>> pos_data(spec.num_chromosomes);
    for (uint32_t c = 0; c < spec.num_chromosomes; ++c) {
        for (uint32_t l = chr[c]; l < chr[c + 1]; ++l) {
            PosData pd{pos[l], std::vector<uint32_t>(rid.begin() + off[l], rid.begin() + off[l + 1]), {}};
            pd.group_ids_bases.assign(idb.begin() + off[l], idb.begin() + off[l + 1]);
            pos_data[c].push_back(std::move(pd));
        }
    }
    std::vector<uint32_t> identity(spec.num_cells);
    for (uint32_t i = 0; i < spec.num_cells; ++i) identity[i] = i;
    double first = 0, best = 1e30, sum = 0;
    unsigned long long hash = 0;
    for (int k = 0; k < 1 + repeats; ++k) {
        const double t0 = now_s();
        Mat m = secedo_amd::computeSimilarityMatrix<Mat, PosData>(pos_data, spec.num_cells, 1000, identity, 0.01, 0.5,
                                                                  0.01, 8, "", "ADD_MIN");
        const double dt = now_s() - t0;
        if (k == 0) first = dt;
        else best = dt < best ? dt : best;
        sum = 0;
        for (uint32_t i = 0; i < spec.num_cells; i += 97) sum += m(i, (i * 31 + 7) % spec.num_cells);
        if (k == repeats) {  // every bit of the matrix (FNV-1a over its 8-byte words)
            hash = 0xcbf29ce484222325ull;
            for (uint32_t i = 0; i < spec.num_cells; ++i)
                for (uint32_t j = 0; j < spec.num_cells; ++j) {
                    unsigned long long w;
                    const double v = m(i, j);
                    std::memcpy(&w, &v, 8);
                    hash = (hash ^ w) * 0x100000001b3ull;
                }
        }
    }
    int ids[16];
    const int n_dev = secedo_simmat_get_devices(ids, 16);
    std::printf("{\"cells\": %u, \"loci\": %llu, \"entries\": %llu, \"first_call_s\": %.6f, \"repeated_call_s\": %.6f, "
                "\"matrix_bytes\": %llu, \"checksum\": %.12g, \"matrix_hash\": \"%016llx\", \"devices\": %d}\n",
                spec.num_cells, (unsigned long long)L, (unsigned long long)E, first, best,
                (unsigned long long)spec.num_cells * spec.num_cells * 8ull, sum, hash, n_dev);
    return 0;
}

int main(int argc, char **argv) {
    // --gpus N | --gpus a,b,c in front of everything else: the devices the (unchanged) computeSimilarityMatrix()
    // call spreads its matrix over, handed to the library the way a deployment does it -- the environment's
    // SECEDO_GPUS (secedo_simmat_set_devices() is the programmatic form, tests/test_gpu_multi_device.py)
    if (argc >= 3 && std::strcmp(argv[1], "--gpus") == 0) {
        setenv("SECEDO_GPUS", argv[2], 1);
        argv[2] = argv[0];
        argv += 2;
        argc -= 2;
    }
    if (argc >= 7 && std::strcmp(argv[1], "--synth") == 0) {
        try {
            return run_synth(argc, argv);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "error: %s\n", e.what());
            return 4;
        }
    }
    if (argc != 7 && argc != 8) {
        std::fprintf(stderr, "usage: %s pileup.bin num_cells mfl num_threads normalization out.f64 [consumers|nodata]\n",
                     argv[0]);
        return 2;
    }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<PosData> chromosome;
    while (true) {
        uint32_t position;
        uint16_t coverage;
        if (!f.read(reinterpret_cast<char *>(&position), 4)) break;
        f.read(reinterpret_cast<char *>(&coverage), 2);
        PosData pd{position, std::vector<uint32_t>(coverage), std::vector<uint16_t>(coverage)};
        f.read(reinterpret_cast<char *>(pd.read_ids.data()), coverage * 4);
        f.read(reinterpret_cast<char *>(pd.group_ids_bases.data()), coverage * 2);
        chromosome.push_back(std::move(pd));
    }
    const uint32_t n = static_cast<uint32_t>(std::atoi(argv[2]));
    const uint32_t mfl = static_cast<uint32_t>(std::atoi(argv[3])), threads = static_cast<uint32_t>(std::atoi(argv[4]));
    std::vector<uint32_t> identity(n);
    for (uint32_t i = 0; i < n; ++i) identity[i] = i;
    const bool consumers = argc == 8 && std::strcmp(argv[7], "consumers") == 0;
    const bool nodata = argc == 8 && std::strcmp(argv[7], "nodata") == 0;
    try {
        std::ofstream out(argv[6], std::ios::binary);
        if (nodata) {
            MatNoData m = secedo_amd::computeSimilarityMatrix<MatNoData, PosData>({chromosome}, n, mfl, identity, 0.01, 0.5,
                                                                                  0.01, threads, "", argv[5]);
            out.write(reinterpret_cast<const char *>(m.elements().data()), sizeof(double) * n * n);
            return 0;
        }
        Mat m = secedo_amd::computeSimilarityMatrix<Mat, PosData>({chromosome}, n, mfl, identity, 0.01, 0.5, 0.01,
                                                                  threads, "", argv[5]);
        out.write(reinterpret_cast<const char *>(m.data()), sizeof(double) * n * n);
        if (consumers) {
            std::vector<double> values, vectors;
            secedo_amd::smallest_eigenpairs(m, 7, 7, &values, &vectors);
            std::vector<double> prob(n);
            for (uint32_t i = 0; i < n; ++i) prob[i] = vectors[static_cast<size_t>(n) + i] >= 0 ? 0.9 : 0.1;
            secedo_amd::expectation_maximization<PosData>({chromosome}, identity, 1, 1e-3, &prob);
            out.write(reinterpret_cast<const char *>(values.data()), sizeof(double) * values.size());
            out.write(reinterpret_cast<const char *>(vectors.data()), sizeof(double) * vectors.size());
            out.write(reinterpret_cast<const char *>(prob.data()), sizeof(double) * prob.size());
        }
    } catch (const std::logic_error &e) {
        std::fprintf(stderr, "logic_error: %s\n", e.what());
        return 3;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 4;
    }
    return 0;
}
