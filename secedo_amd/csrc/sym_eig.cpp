// sym_eig.cpp -- eigen-decomposition of a small dense symmetric matrix on the host: the
// Rayleigh-Ritz step of the block Krylov iteration in spectral_api.cpp (a few hundred rows at most;
// the N x N work stays on the GPU). Householder reduction to tridiagonal form, then the implicit
// QL iteration with Wilkinson shifts; the transformations are accumulated.
#include "sym_eig.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace secedo {

namespace {

// A (n x n, row-major, symmetric) -> tridiagonal (d, e) with A = Q T Q^T; Q^T overwrites A
void tridiagonalise(int n, std::vector<double> &a, std::vector<double> &d, std::vector<double> &e) {
    std::vector<double> v(n), p(n), q((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) q[(size_t)i * n + i] = 1.0;
    for (int k = 0; k + 2 < n; ++k) {
        // Householder vector for column k below the subdiagonal
        double norm2 = 0.0;
        for (int i = k + 1; i < n; ++i) norm2 += a[(size_t)i * n + k] * a[(size_t)i * n + k];
        const double x0 = a[(size_t)(k + 1) * n + k];
        double tail2 = norm2 - x0 * x0;
        if (tail2 <= 0.0) continue;  // already tridiagonal in this column
        const double norm = std::sqrt(norm2);
        const double alpha = x0 > 0.0 ? -norm : norm;
        double vnorm2 = 0.0;
        for (int i = k + 1; i < n; ++i) {
            v[i] = a[(size_t)i * n + k];
            if (i == k + 1) v[i] -= alpha;
            vnorm2 += v[i] * v[i];
        }
        if (vnorm2 == 0.0) continue;
        const double inv = 1.0 / std::sqrt(vnorm2);
        for (int i = k + 1; i < n; ++i) v[i] *= inv;
        // trailing block B = A[k+1:, k+1:]: B <- H B H, H = I - 2 v v^T
        for (int i = k + 1; i < n; ++i) {
            double s = 0.0;
            for (int j = k + 1; j < n; ++j) s += a[(size_t)i * n + j] * v[j];
            p[i] = s;
        }
        double kappa = 0.0;
        for (int i = k + 1; i < n; ++i) kappa += v[i] * p[i];
        for (int i = k + 1; i < n; ++i) p[i] -= kappa * v[i];  // w
        for (int i = k + 1; i < n; ++i) {
            for (int j = k + 1; j < n; ++j) a[(size_t)i * n + j] -= 2.0 * (v[i] * p[j] + p[i] * v[j]);
        }
        a[(size_t)(k + 1) * n + k] = alpha;
        a[(size_t)k * n + k + 1] = alpha;
        for (int i = k + 2; i < n; ++i) {
            a[(size_t)i * n + k] = 0.0;
            a[(size_t)k * n + i] = 0.0;
        }
        // Q <- Q H, kept transposed (q holds Q^T: contiguous rows for the rotations that follow):
        // Q^T <- H Q^T = Q^T - 2 v (v^T Q^T), rows k+1.. only
        std::fill(p.begin(), p.end(), 0.0);
        for (int j = k + 1; j < n; ++j) {
            const double vj = v[j];
            const double *row = &q[(size_t)j * n];
            for (int i = 0; i < n; ++i) p[i] += vj * row[i];
        }
        for (int j = k + 1; j < n; ++j) {
            const double vj = 2.0 * v[j];
            double *row = &q[(size_t)j * n];
            for (int i = 0; i < n; ++i) row[i] -= vj * p[i];
        }
    }
    d.assign(n, 0.0);
    e.assign(n, 0.0);
    for (int i = 0; i < n; ++i) {
        d[i] = a[(size_t)i * n + i];
        if (i + 1 < n) e[i] = a[(size_t)(i + 1) * n + i];
    }
    a.swap(q);
}

// implicit QL with Wilkinson shifts on (d, e); rotations accumulated into z (ROWS = vectors)
bool ql_implicit(int n, std::vector<double> &d, std::vector<double> &e, std::vector<double> &z) {
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        while (true) {
            int m = l;
            for (; m + 1 < n; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m == l) break;
            if (++iter > 200) return false;
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = std::hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            for (; i >= l; --i) {
                double f = s * e[i];
                const double b = c * e[i];
                r = std::hypot(f, g);
                e[i + 1] = r;
                if (r == 0.0) {
                    d[i + 1] -= p;
                    e[m] = 0.0;
                    break;
                }
                s = f / r;
                c = g / r;
                g = d[i + 1] - p;
                r = (d[i] - g) * s + 2.0 * c * b;
                p = s * r;
                d[i + 1] = g + p;
                g = c * r - b;
                double *zi = &z[(size_t)i * n], *zj = &z[(size_t)(i + 1) * n];
                for (int k = 0; k < n; ++k) {
                    const double fk = zj[k];
                    zj[k] = s * zi[k] + c * fk;
                    zi[k] = c * zi[k] - s * fk;
                }
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
        }
    }
    return true;
}

}  // namespace

bool sym_eig(int n, const std::vector<double> &a_in, std::vector<double> &evals, std::vector<double> &evecs) {
    if (n <= 0) {
        evals.clear();
        evecs.clear();
        return true;
    }
    std::vector<double> a((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) a[(size_t)i * n + j] = 0.5 * (a_in[(size_t)i * n + j] + a_in[(size_t)j * n + i]);
    std::vector<double> d, e;
    tridiagonalise(n, a, d, e);
    if (!ql_implicit(n, d, e, a)) return false;
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] < d[y]; });
    evals.resize(n);
    evecs.assign((size_t)n * n, 0.0);
    for (int k = 0; k < n; ++k) {
        evals[k] = d[order[k]];
        for (int i = 0; i < n; ++i) evecs[(size_t)i * n + k] = a[(size_t)order[k] * n + i];
    }
    return true;
}

}  // namespace secedo
