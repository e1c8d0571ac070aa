// sym_eig.cpp -- eigen-decomposition of a small dense symmetric matrix on the host: the
// Rayleigh-Ritz step of the block Krylov iteration in spectral_api.cpp (a few hundred rows at most;
// the N x N work stays on the GPU). Householder reduction to tridiagonal form, then the implicit
// QL iteration with Wilkinson shifts; the transformations are accumulated.
#include "sym_eig.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

namespace secedo {

bool sym_eig(int n, const std::vector<double> &a_in, std::vector<double> &evals, std::vector<double> &evecs);

namespace {

// sqrt(a^2 + b^2); std::hypot guards against overflow at ten times the cost, which only matters far
// outside the range of a projected operator with norm <= 1
inline double pythag(double a, double b) {
    const double m = std::fmax(std::fabs(a), std::fabs(b));
    if (m > 1e-140 && m < 1e140) return std::sqrt(a * a + b * b);
    return std::hypot(a, b);
}

// Inner products and updates of the Rayleigh-Ritz step (sym_eig_top is on the critical path of every
// restart cycle). Eight independent partial sums: the order of the additions is fixed by the code, so
// the result does not depend on the machine, and the compiler may keep the sums in vector registers
// (AVX2 clone picked at load time where the CPU has it).
#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
#define SECEDO_SIMD_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define SECEDO_SIMD_CLONES
#endif

SECEDO_SIMD_CLONES double dot8(const double *x, const double *y, int n) {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int u = 0; u < 8; ++u) s[u] += x[i + u] * y[i + u];
    for (int u = 0; i < n; ++i, ++u) s[u] += x[i] * y[i];
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// y += alpha * x
SECEDO_SIMD_CLONES void axpy(double alpha, const double *x, double *y, int n) {
    for (int i = 0; i < n; ++i) y[i] += alpha * x[i];
}

// row -= a * p + b * v   (the symmetric rank-2 update of one row of the lower triangle)
SECEDO_SIMD_CLONES void rank2_row(double a, const double *p, double b, const double *v, double *row, int n) {
    for (int i = 0; i < n; ++i) row[i] -= a * p[i] + b * v[i];
}

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
            double r = pythag(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            for (; i >= l; --i) {
                double f = s * e[i];
                const double b = c * e[i];
                r = pythag(f, g);
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

// Householder reduction that keeps the reflectors (column k of `vs` = v_k, unit norm, zero up to row
// k): Q = H_0 H_1 ... is applied to single vectors afterwards instead of being formed.
void tridiagonalise_keep(int n, std::vector<double> &a, std::vector<double> &d, std::vector<double> &e,
                         std::vector<double> &vs, std::vector<char> &used) {
    std::vector<double> v(n), p(n);
    vs.assign((size_t)n * n, 0.0);
    used.assign(n, 0);
    for (int k = 0; k + 2 < n; ++k) {
        double norm2 = 0.0;  // column k below the diagonal, from the lower triangle
        for (int i = k + 1; i < n; ++i) norm2 += a[(size_t)i * n + k] * a[(size_t)i * n + k];
        const double x0 = a[(size_t)(k + 1) * n + k];
        if (norm2 - x0 * x0 <= 0.0) continue;
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
        // only the lower triangle of the trailing block is kept up to date: p = B v from it
        for (int i = k + 1; i < n; ++i) p[i] = 0.0;
        for (int i = k + 1; i < n; ++i) {
            const double *row = &a[(size_t)i * n];
            const int len = i - (k + 1);
            p[i] += row[i] * v[i] + dot8(row + k + 1, &v[k + 1], len);
            axpy(v[i], row + k + 1, &p[k + 1], len);
        }
        const double kappa = dot8(&v[k + 1], &p[k + 1], n - (k + 1));
        axpy(-kappa, &v[k + 1], &p[k + 1], n - (k + 1));
        for (int i = k + 1; i < n; ++i)
            rank2_row(2.0 * v[i], &p[k + 1], 2.0 * p[i], &v[k + 1], &a[(size_t)i * n + k + 1], i - k);
        a[(size_t)(k + 1) * n + k] = alpha;
        for (int i = k + 2; i < n; ++i) a[(size_t)i * n + k] = 0.0;
        for (int i = k + 1; i < n; ++i) vs[(size_t)k * n + i] = v[i];  // reflector k, stored as a row
        used[k] = 1;
    }
    d.assign(n, 0.0);
    e.assign(n, 0.0);
    for (int i = 0; i < n; ++i) {
        d[i] = a[(size_t)i * n + i];
        if (i + 1 < n) e[i] = a[(size_t)(i + 1) * n + i];
    }
}

// One row of the Sturm recurrence for k shifts at once: q <- d_i - shift - e_{i-1}^2 / q, negative pivots
// counted (a vanishing pivot is pushed to -pivmin, as in LAPACK's dstebz). Counts kept as doubles so that
// the loop over the shifts is one vector loop.
SECEDO_SIMD_CLONES void sturm_row(int k, double di, double e2, double pivmin, const double *shift, double *q,
                                  double *negatives) {
    for (int j = 0; j < k; ++j) {
        double t = q[j];
        t = std::fabs(t) < pivmin ? -pivmin : t;
        const double v = di - shift[j] - e2 / t;
        q[j] = v;
        negatives[j] += v < 0.0 ? 1.0 : 0.0;
    }
}

// The k largest eigenvalues of the tridiagonal (d, e), descending, by bisection on the Sturm count --
// all k intervals advance together. Each halving costs one pass over the matrix for all k values,
// against the O(n^2) rotations QL spends on the n - k values nobody reads.
void top_values_bisect(int n, const std::vector<double> &d, const std::vector<double> &e, int k, double *out) {
    double gl = d[0], gu = d[0], emax2 = 0.0;
    std::vector<double> e2(n, 0.0);
    for (int i = 0; i < n; ++i) {
        const double r = (i ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(e[i]) : 0.0);
        gl = std::min(gl, d[i] - r);
        gu = std::max(gu, d[i] + r);
        if (i + 1 < n) {
            e2[i] = e[i] * e[i];
            emax2 = std::max(emax2, e2[i]);
        }
    }
    const double eps = 2.220446049250313e-16, pivmin = 2.2250738585072014e-308 * std::max(1.0, emax2);
    const double tn = std::max(std::fabs(gl), std::fabs(gu));
    gl -= 2.0 * tn * eps * n + 2.0 * pivmin;
    gu += 2.0 * tn * eps * n + 2.0 * pivmin;
    std::vector<double> lo(k, gl), hi(k, gu), mid(k), q(k), neg(k);
    for (int iter = 0; iter < 80; ++iter) {
        bool open = false;
        for (int j = 0; j < k; ++j) {
            mid[j] = 0.5 * (lo[j] + hi[j]);
            open = open || hi[j] - lo[j] > 2.0 * eps * std::max(std::fabs(lo[j]), std::fabs(hi[j])) + 2.0 * pivmin;
        }
        if (!open) break;
        for (int j = 0; j < k; ++j) {
            q[j] = d[0] - mid[j];
            neg[j] = q[j] < 0.0 ? 1.0 : 0.0;
        }
        for (int i = 1; i < n; ++i) sturm_row(k, d[i], e2[i - 1], pivmin, mid.data(), q.data(), neg.data());
        // the j-th largest eigenvalue lies below x exactly when at least n - j eigenvalues do
        for (int j = 0; j < k; ++j) {
            if (neg[j] >= (double)(n - j)) hi[j] = mid[j];
            else lo[j] = mid[j];
        }
    }
    for (int j = 0; j < k; ++j) out[j] = 0.5 * (lo[j] + hi[j]);
}

// eigenvalues of the tridiagonal (d, e) by the implicit QL iteration, no vectors
bool ql_values(int n, std::vector<double> d, std::vector<double> e, std::vector<double> &out) {
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
            double r = pythag(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            for (; i >= l; --i) {
                const double f = s * e[i], b = c * e[i];
                r = pythag(f, g);
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
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
        }
    }
    out = d;
    std::sort(out.begin(), out.end());
    return true;
}

// One eigenvector of the tridiagonal (d, e) for the eigenvalue approximation lambda by inverse
// iteration: LU of T - lambda I with partial pivoting (a second super-diagonal appears), a few
// solves from a fixed pseudo-random start, orthogonalised against `against` (the vectors already
// found for eigenvalues close to lambda) before each normalisation.
bool inverse_iteration(int n, const std::vector<double> &d, const std::vector<double> &e, double lambda, double tnorm,
                       const std::vector<const double *> &against, unsigned seed, double *x) {
    std::vector<double> u0(n), u1(n, 0.0), u2(n, 0.0), lmul(n, 0.0);
    std::vector<char> swapped(n, 0);
    const double tiny = 2.3e-16 * (tnorm > 0.0 ? tnorm : 1.0);
    // elimination: row i has (u0[i], u1[i], u2[i]) on columns (i, i+1, i+2) after the step
    double di = d[0] - lambda, ei = n > 1 ? e[0] : 0.0, fi = 0.0;  // current row i
    for (int i = 0; i + 1 < n; ++i) {
        const double sub = e[i];                                        // row i+1: (sub, d[i+1]-lambda, e[i+1])
        const double dn = d[i + 1] - lambda, en = (i + 2 < n) ? e[i + 1] : 0.0;
        if (std::fabs(sub) > std::fabs(di)) {  // swap rows i and i+1
            swapped[i] = 1;
            lmul[i] = di / sub;
            u0[i] = sub;
            u1[i] = dn;
            u2[i] = en;
            const double ndi = ei - lmul[i] * dn, nei = fi - lmul[i] * en;
            di = ndi;
            ei = nei;
            fi = 0.0;
        } else {
            const double piv = std::fabs(di) < tiny ? (di < 0 ? -tiny : tiny) : di;
            lmul[i] = sub / piv;
            u0[i] = piv;
            u1[i] = ei;
            u2[i] = fi;
            di = dn - lmul[i] * ei;
            ei = en - lmul[i] * fi;
            fi = 0.0;
        }
    }
    u0[n - 1] = std::fabs(di) < tiny ? (di < 0 ? -tiny : tiny) : di;
    unsigned long long state = 0x9E3779B97F4A7C15ull * (seed + 1);
    for (int i = 0; i < n; ++i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        x[i] = 0.5 + (double)(state >> 11) * (1.0 / 9007199254740992.0);  // (0.5, 1.5): no accidental zero start
    }
    std::vector<double> y(n);
    for (int it = 0; it < 6; ++it) {
        // forward: apply the row operations to x
        for (int i = 0; i < n; ++i) y[i] = x[i];
        for (int i = 0; i + 1 < n; ++i) {
            if (swapped[i]) std::swap(y[i], y[i + 1]);
            y[i + 1] -= lmul[i] * y[i];
        }
        // back substitution with the upper triangle (three diagonals)
        for (int i = n - 1; i >= 0; --i) {
            double v = y[i];
            if (i + 1 < n) v -= u1[i] * y[i + 1];
            if (i + 2 < n) v -= u2[i] * y[i + 2];
            y[i] = v / u0[i];
        }
        for (const double *q : against) {
            double dot = 0.0;
            for (int i = 0; i < n; ++i) dot += q[i] * y[i];
            for (int i = 0; i < n; ++i) y[i] -= dot * q[i];
        }
        double norm = 0.0;
        for (int i = 0; i < n; ++i) norm += y[i] * y[i];
        norm = std::sqrt(norm);
        if (!(norm > 0.0) || !std::isfinite(norm)) return false;
        for (int i = 0; i < n; ++i) x[i] = y[i] / norm;
        // converged when T x - lambda x is at rounding level
        double res = 0.0;
        for (int i = 0; i < n; ++i) {
            double r = (d[i] - lambda) * x[i];
            if (i > 0) r += e[i - 1] * x[i - 1];
            if (i + 1 < n) r += e[i] * x[i + 1];
            res = std::max(res, std::fabs(r));
        }
        if (it >= 1 && res <= 64.0 * tiny) return true;
    }
    return false;
}

}  // namespace

bool sym_eig_top(int n, const std::vector<double> &a_in, int k, std::vector<double> &evals,
                 std::vector<double> &top_vecs, bool top_values_only) {
    if (k > n) k = n;
    if (n <= 2 || k <= 0) {  // nothing to gain: the full decomposition
        std::vector<double> z;
        if (!sym_eig(n, a_in, evals, z)) return false;
        top_vecs.assign((size_t)n * std::max(k, 0), 0.0);
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < n; ++i) top_vecs[(size_t)i * k + j] = z[(size_t)i * n + (n - 1 - j)];
        return true;
    }
    std::vector<double> a((size_t)n * n);
    double tnorm = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[(size_t)i * n + j] = 0.5 * (a_in[(size_t)i * n + j] + a_in[(size_t)j * n + i]);
            tnorm = std::max(tnorm, std::fabs(a[(size_t)i * n + j]));
        }
    const std::vector<double> a_sym = a;
    std::vector<double> d, e, vs;
    std::vector<char> used;
    tridiagonalise_keep(n, a, d, e, vs, used);
    if (top_values_only) {
        evals.assign(n, std::numeric_limits<double>::quiet_NaN());
        std::vector<double> topv(k);
        top_values_bisect(n, d, e, k, topv.data());
        for (int j = 0; j < k; ++j) evals[n - 1 - j] = topv[j];
    } else if (!ql_values(n, d, e, evals)) {
        return false;
    }
    double t1 = 0.0;  // 1-norm of the tridiagonal: the scale of its rounding errors
    for (int i = 0; i < n; ++i) t1 = std::max(t1, std::fabs(d[i]) + (i ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(e[i]) : 0.0));
    std::vector<double> tv((size_t)k * n);  // eigenvectors of the tridiagonal, one per row
    std::vector<double> lam(k);
    for (int j = 0; j < k; ++j) {
        lam[j] = evals[n - 1 - j];
        // eigenvalues closer than 1e-3 of the norm form a cluster: orthogonalise inside it, and nudge
        // coinciding shifts apart so that the solves do not all return the same direction
        std::vector<const double *> against;
        double shift = lam[j];
        for (int q = 0; q < j; ++q) {
            if (std::fabs(lam[q] - lam[j]) <= 1e-3 * t1) against.push_back(&tv[(size_t)q * n]);
        }
        if (j > 0 && lam[j - 1] - shift < 10.0 * 2.3e-16 * t1) shift = lam[j - 1] - 10.0 * 2.3e-16 * t1 * (double)against.size();
        if (!inverse_iteration(n, d, e, shift, t1, against, (unsigned)j, &tv[(size_t)j * n])) {
            std::vector<double> z;  // rare: fall back to the full decomposition
            if (!sym_eig(n, a_in, evals, z)) return false;
            top_vecs.assign((size_t)n * k, 0.0);
            for (int jj = 0; jj < k; ++jj)
                for (int i = 0; i < n; ++i) top_vecs[(size_t)i * k + jj] = z[(size_t)i * n + (n - 1 - jj)];
            return true;
        }
    }
    // back-transformation: eigenvector of A = H_0 H_1 ... H_{n-3} x
    for (int j = 0; j < k; ++j) {
        double *x = &tv[(size_t)j * n];
        for (int r = n - 3; r >= 0; --r) {
            if (!used[r]) continue;
            const double *v = &vs[(size_t)r * n];
            const int len = n - (r + 1);
            axpy(-2.0 * dot8(v + r + 1, x + r + 1, len), v + r + 1, x + r + 1, len);
        }
    }
    // accept only what A itself confirms
    for (int j = 0; j < k; ++j) {
        const double *x = &tv[(size_t)j * n];
        double res = 0.0;
        for (int i = 0; i < n; ++i) {
            const double s = dot8(&a_sym[(size_t)i * n], x, n);
            res = std::max(res, std::fabs(s - lam[j] * x[i]));
        }
        if (!(res <= 1e-11 * (tnorm > 0.0 ? tnorm * n : 1.0))) {
            std::vector<double> z;
            if (!sym_eig(n, a_in, evals, z)) return false;
            top_vecs.assign((size_t)n * k, 0.0);
            for (int jj = 0; jj < k; ++jj)
                for (int i = 0; i < n; ++i) top_vecs[(size_t)i * k + jj] = z[(size_t)i * n + (n - 1 - jj)];
            return true;
        }
    }
    top_vecs.assign((size_t)n * k, 0.0);
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < n; ++i) top_vecs[(size_t)i * k + j] = tv[(size_t)j * n + i];
    return true;
}

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
