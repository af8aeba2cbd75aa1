// cache.hpp -- host-side (cold path, once per problem family) computation of the TinyCache:
// the infinite-horizon Riccati recursion of the reference's tiny_precompute_and_set_cache
// (src/tinympc/tiny_api.cpp:307-381) and the work->Q / work->R bookkeeping of tiny_setup
// (tiny_api.cpp:117-118,136).  Results feed every kernel launch, so the reference's quirks are
// reproduced on purpose:
//   * rho enters Q/R TWICE on the cache path (tiny_api.cpp:117-118 then :317-318) but once in
//     work->Q / work->R;
//   * the recursion breaks BEFORE Ptp1 = Pinf (tiny_api.cpp:340-348), so the returned Pinf is one
//     Riccati step ahead of the P that produced the returned Kinf;
//   * products associate left to right as the Eigen expressions do.
// Small dense column-major matrices; nothing here is performance critical.
#pragma once
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

namespace tinympc_amd {

struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() = default;
    Mat(int rows, int cols) : r(rows), c(cols), a((size_t)rows * cols, 0.0) {}
    Mat(int rows, int cols, const double* src) : r(rows), c(cols), a(src, src + (size_t)rows * cols) {}
    double& operator()(int i, int j) { return a[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
    static Mat diag(const std::vector<double>& d) {
        Mat m((int)d.size(), (int)d.size());
        for (int i = 0; i < (int)d.size(); ++i) m(i, i) = d[i];
        return m;
    }
};

inline Mat operator*(const Mat& x, const Mat& y) {
    Mat z(x.r, y.c);
    for (int j = 0; j < y.c; ++j)
        for (int i = 0; i < x.r; ++i) {
            double s = 0.0;
            for (int k = 0; k < x.c; ++k) s += x(i, k) * y(k, j);
            z(i, j) = s;
        }
    return z;
}
inline Mat operator+(const Mat& x, const Mat& y) {
    Mat z(x.r, x.c);
    for (size_t i = 0; i < z.a.size(); ++i) z.a[i] = x.a[i] + y.a[i];
    return z;
}
inline Mat operator-(const Mat& x, const Mat& y) {
    Mat z(x.r, x.c);
    for (size_t i = 0; i < z.a.size(); ++i) z.a[i] = x.a[i] - y.a[i];
    return z;
}
inline Mat transpose(const Mat& x) {
    Mat z(x.c, x.r);
    for (int j = 0; j < x.c; ++j)
        for (int i = 0; i < x.r; ++i) z(j, i) = x(i, j);
    return z;
}
inline double max_abs_diff(const Mat& x, const Mat& y) {
    double m = 0.0;
    for (size_t i = 0; i < x.a.size(); ++i) m = std::fmax(m, std::fabs(x.a[i] - y.a[i]));
    return m;
}

// Inverse through LU with partial pivoting (what Eigen's dynamic-size .inverse() does).
inline bool invert(const Mat& m, Mat* out) {
    const int n = m.r;
    Mat lu = m;
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(lu(i, k)) > std::fabs(lu(p, k))) p = i;
        if (lu(p, k) == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(lu(k, j), lu(p, j));
            std::swap(perm[k], perm[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            lu(i, k) /= lu(k, k);
            for (int j = k + 1; j < n; ++j) lu(i, j) -= lu(i, k) * lu(k, j);
        }
    }
    *out = Mat(n, n);
    for (int col = 0; col < n; ++col) {
        std::vector<double> x(n);
        for (int i = 0; i < n; ++i) {
            double s = (perm[i] == col) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s -= lu(i, j) * x[j];
            x[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = x[i];
            for (int j = i + 1; j < n; ++j) s -= lu(i, j) * x[j];
            x[i] = s / lu(i, i);
        }
        for (int i = 0; i < n; ++i) (*out)(i, col) = x[i];
    }
    return true;
}

struct Cache {                    // TinyCache, types.hpp:43-59 (hot-path members)
    double rho = 0.0;
    Mat Kinf, Pinf, Quu_inv, AmBKt, APf, BPf;
    int riccati_iters = 0;
    bool riccati_converged = false;
};

// Q / R are the (dense) matrices handed to tiny_precompute_and_set_cache, i.e. ALREADY
// diag(work->Q) / diag(work->R) (user + rho) when called from tiny_setup; rho is added again
// here (:317-318).
inline bool precompute_cache(const Mat& A, const Mat& B, const Mat& f, const Mat& Q, const Mat& R, double rho,
                             Cache* c) {
    const int nx = A.r, nu = B.c;
    Mat Q1 = Q, R1 = R;
    for (int i = 0; i < nx; ++i) Q1(i, i) += rho;
    for (int i = 0; i < nu; ++i) R1(i, i) += rho;
    const Mat At = transpose(A), Bt = transpose(B);
    Mat Ktp1(nu, nx), Ptp1 = Mat::diag(std::vector<double>(nx, rho));
    Mat Kinf(nu, nx), Pinf(nx, nx), Ginv;
    c->riccati_iters = 1000;
    for (int i = 0; i < 1000; ++i) {
        if (!invert(R1 + (Bt * Ptp1) * B, &Ginv)) return false;
        Kinf = ((Ginv * Bt) * Ptp1) * A;                         // :337
        Pinf = Q1 + (At * Ptp1) * (A - B * Kinf);                // :338
        if (max_abs_diff(Kinf, Ktp1) < 1e-5) { c->riccati_iters = i + 1; c->riccati_converged = true; break; }   // :340-346
        Ktp1 = Kinf;
        Ptp1 = Pinf;
    }
    if (!invert(R1 + (Bt * Pinf) * B, &c->Quu_inv)) return false;   // :352
    c->AmBKt = transpose(A - B * Kinf);                              // :353
    c->APf = (c->AmBKt * Pinf) * f;                                  // :356
    c->BPf = (Bt * Pinf) * f;                                        // :357
    c->Kinf = Kinf;
    c->Pinf = Pinf;
    c->rho = rho;
    return true;
}

}  // namespace tinympc_amd
