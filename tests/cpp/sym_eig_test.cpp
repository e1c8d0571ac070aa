// Host test of secedo_amd/csrc/sym_eig.cpp (the Rayleigh-Ritz eigensolver of the spectral step):
// residuals and orthonormality of sym_eig and sym_eig_top on random, clustered and degenerate
// symmetric matrices. Prints "ok" and exits 0, or a diagnostic and 1.
#include "sym_eig.hpp"

#include <cmath>
#include <cstdio>
#include <random>

static int check(int n, int mode) {
    std::mt19937_64 rng(n * 7 + mode);
    std::normal_distribution<double> nd;
    std::vector<double> a((size_t)n * n, 0.0);
    if (mode == 0) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) a[i * n + j] = a[j * n + i] = nd(rng);
    } else if (mode == 1) {  // Q diag Q^T with a dense cluster of eigenvalues below three separated ones
        std::vector<double> q((size_t)n * n);
        for (auto &v : q) v = nd(rng);
        for (int c = 0; c < n; ++c) {
            for (int p = 0; p < c; ++p) {
                double d = 0;
                for (int i = 0; i < n; ++i) d += q[i * n + c] * q[i * n + p];
                for (int i = 0; i < n; ++i) q[i * n + c] -= d * q[i * n + p];
            }
            double nn = 0;
            for (int i = 0; i < n; ++i) nn += q[i * n + c] * q[i * n + c];
            nn = std::sqrt(nn);
            for (int i = 0; i < n; ++i) q[i * n + c] /= nn;
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0;
                for (int c = 0; c < n; ++c) {
                    const double lam = c < 3 ? 1.0 - 0.1 * c : 0.5 + 1e-6 * c + (c % 5 == 0 ? 0.0 : 1e-9 * c);
                    s += q[i * n + c] * lam * q[j * n + c];
                }
                a[i * n + j] = s;
            }
    } else {  // decoupled blocks, exact multiplicities, "dropped column" rows (-1 on the diagonal)
        for (int i = 0; i < n; ++i) a[i * n + i] = (i % 4 == 0) ? -1.0 : 0.5;
        for (int i = 1; i < n; ++i)
            if (i % 4 && (i - 1) % 4) a[i * n + i - 1] = a[(i - 1) * n + i] = 0.1;
    }
    const int k = n < 32 ? n : 32;
    std::vector<double> ev, tv, ev2, z;
    if (!secedo::sym_eig_top(n, a, k, ev, tv) || !secedo::sym_eig(n, a, ev2, z)) return 1;
    double res = 0, orth = 0, dv = 0, res2 = 0;
    for (int i = 0; i < n; ++i) dv = std::fmax(dv, std::fabs(ev[i] - ev2[i]));
    for (int j = 0; j < k; ++j) {
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int c = 0; c < n; ++c) s += a[i * n + c] * tv[c * k + j];
            res = std::fmax(res, std::fabs(s - ev[n - 1 - j] * tv[i * k + j]));
        }
        for (int l = 0; l < k; ++l) {
            double s = 0;
            for (int i = 0; i < n; ++i) s += tv[i * k + j] * tv[i * k + l];
            orth = std::fmax(orth, std::fabs(s - (j == l)));
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int c = 0; c < n; ++c) s += a[i * n + c] * z[c * n + j];
            res2 = std::fmax(res2, std::fabs(s - ev2[j] * z[i * n + j]));
        }
    // the variant that finds only the k largest values (bisection): same values, same quality of vectors
    std::vector<double> ev3, tv3;
    if (!secedo::sym_eig_top(n, a, k, ev3, tv3, true)) return 1;
    for (int j = 0; j < k; ++j) {
        dv = std::fmax(dv, std::fabs(ev3[n - 1 - j] - ev2[n - 1 - j]));
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int c = 0; c < n; ++c) s += a[i * n + c] * tv3[c * k + j];
            res = std::fmax(res, std::fabs(s - ev3[n - 1 - j] * tv3[i * k + j]));
        }
        for (int l = 0; l < k; ++l) {
            double s = 0;
            for (int i = 0; i < n; ++i) s += tv3[i * k + j] * tv3[i * k + l];
            orth = std::fmax(orth, std::fabs(s - (j == l)));
        }
    }
    const double tol = 1e-12 * n;
    if (dv > tol || res > tol || orth > tol || res2 > tol) {
        std::printf("n=%d mode=%d: values %.2e residual %.2e orthogonality %.2e full residual %.2e\n", n, mode, dv, res,
                    orth, res2);
        return 1;
    }
    return 0;
}

int main() {
    int bad = 0;
    for (int n : {1, 2, 3, 5, 17, 33, 64, 192})
        for (int mode = 0; mode < 3; ++mode) bad += check(n, mode);
    if (bad) return 1;
    std::printf("ok\n");
    return 0;
}
