// riccati_kernel.hip.h -- batched cache precompute on the GPU for heterogeneous problem families
// (SURVEY.md section 8(f) rank 3): every instance has its own (A, B, f, Q, R, rho).
//
// riccati_kernel : tiny_precompute_and_set_cache (reference src/tinympc/tiny_api.cpp:307-381) for `batch`
//                  instances at once -- one wavefront per instance, all matrices in LDS, the <=1000-step
//                  infinite-horizon recursion with the reference's quirks (rho added a second time, break
//                  BEFORE Ptp1 = Pinf, left-to-right products).  Arithmetic is not FMA-contracted and every
//                  dot product runs in the same order as the host code (cache.hpp): identical problem data give
//                  the same Riccati step count and caches equal to ~1e-13 on both paths (tests/test_gpu_hetero.py).
//                  Its epilogue turns the cache into the per-instance lane tables admm_solve_kernel reads (same
//                  layout and the same pre-multiplied Quu_inv B', Quu_inv BPf as batch_tables.hip:build_tables).
#pragma once
#include <hip/hip_runtime.h>

#include "admm_kernel.hip.h"

namespace tinympc_amd {

struct RiccatiArgs {
    const double *A, *B, *f, *Qw, *Rw, *rho;        // [batch][nx*nx], [batch][nx*nu], [batch][nx], [batch][nx], [batch][nu], [batch]
    double *Kinf, *Pinf, *Quu_inv, *AmBKt, *APf, *BPf;
    int* iters;                                      // Riccati steps taken (1000 = not converged)
    double* tabs;                                    // [batch][tab_doubles] lane tables
    int nx, nu, batch;
    // layout of the lane tables: matrices [column k (0 .. tab_cols-1)][tab_lw lanes] at 0 (MB), cols*lw (MF1), 2 cols*lw (MF2), 3 cols*lw
    // (PT), the 16 lane vectors at 4 cols*lw.  One-row kernel (admm_kernel.hip.h TAB_*): cols = lw = 16; tile kernel
    // (tile_kernel.hip.h TileTab<W>): cols = 32, lw = 16 W
    int tab_cols, tab_lw;
};
constexpr int het_tab_doubles(int cols, int lw) { return 4 * cols * lw + 16 * lw; }

// per-instance table = the matrix + vector part of the shared table (bounds / cones / masks stay shared)
enum : int { HET_TAB_DOUBLES = TAB_BOUNDS, VEC_RHO = 8 };

#ifdef TINYMPC_GENERAL_KERNEL_IMPL   // compiled into batch_dispatch.hip only

#pragma clang fp contract(off)

// C(m x n) = X(m x k) * Y(k x n), column-major, sequential k (same order as cache.hpp's operator*)
__device__ __forceinline__ void w_mm(int m, int k, int n, const double* X, const double* Y, double* C, int lane) {
    for (int e = lane; e < m * n; e += 64) {
        const int i = e % m, j = e / m;
        double s = 0.0;
        for (int l = 0; l < k; ++l) s = __dadd_rn(s, __dmul_rn(X[i + m * l], Y[l + k * j]));
        C[e] = s;
    }
    __syncthreads();
}

// inverse by LU with partial pivoting (cache.hpp:invert); n <= 16; executed by lane 0, result in LDS
__device__ __forceinline__ bool w_invert(int n, const double* G, double* Ginv, double* lu, int lane) {
    __shared__ int ok;
    if (lane == 0) {
        int perm[16];
        for (int e = 0; e < n * n; ++e) lu[e] = G[e];
        for (int i = 0; i < n; ++i) perm[i] = i;
        bool good = true;
        for (int k = 0; k < n && good; ++k) {
            int p = k;
            for (int i = k + 1; i < n; ++i)
                if (fabs(lu[i + n * k]) > fabs(lu[p + n * k])) p = i;
            if (lu[p + n * k] == 0.0) { good = false; break; }
            if (p != k) {
                for (int j = 0; j < n; ++j) { const double t = lu[k + n * j]; lu[k + n * j] = lu[p + n * j]; lu[p + n * j] = t; }
                const int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
            }
            for (int i = k + 1; i < n; ++i) {
                lu[i + n * k] = lu[i + n * k] / lu[k + n * k];
                for (int j = k + 1; j < n; ++j) lu[i + n * j] = __dadd_rn(lu[i + n * j], -__dmul_rn(lu[i + n * k], lu[k + n * j]));
            }
        }
        if (good)
            for (int col = 0; col < n; ++col) {
                double x[16];
                for (int i = 0; i < n; ++i) {
                    double s = (perm[i] == col) ? 1.0 : 0.0;
                    for (int j = 0; j < i; ++j) s = __dadd_rn(s, -__dmul_rn(lu[i + n * j], x[j]));
                    x[i] = s;
                }
                for (int i = n - 1; i >= 0; --i) {
                    double s = x[i];
                    for (int j = i + 1; j < n; ++j) s = __dadd_rn(s, -__dmul_rn(lu[i + n * j], x[j]));
                    x[i] = s / lu[i + n * i];
                }
                for (int i = 0; i < n; ++i) Ginv[i + n * col] = x[i];
            }
        ok = good ? 1 : 0;
    }
    __syncthreads();
    return ok != 0;
}

__global__ __launch_bounds__(64) void riccati_kernel(const RiccatiArgs P) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x, nx = P.nx, nu = P.nu;
    const int xx = nx * nx, xu = nx * nu, uu = nu * nu;
    double* A = sm;            double* B = A + xx;      double* At = B + xu;     double* Bt = At + xx;
    double* Pt = Bt + xu;      double* Pn = Pt + xx;    double* K = Pn + xx;     double* Kp = K + xu;
    double* T1 = Kp + xu;      double* T2 = T1 + xu;    double* BtP = T2 + xu;   double* G = BtP + xu;
    double* Gi = G + uu;       double* lu = Gi + uu;    double* AtP = lu + uu;   double* BK = AtP + xx;
    double* T3 = BK + xx;      double* q1 = T3 + xx;    double* r1 = q1 + nx;    double* fv = r1 + nu;
    for (int b = blockIdx.x; b < P.batch; b += gridDim.x) {
        const double rho = P.rho[b];
        for (int e = lane; e < xx; e += 64) { A[e] = P.A[(size_t)b * xx + e]; Pt[e] = 0.0; }
        for (int e = lane; e < xu; e += 64) { B[e] = P.B[(size_t)b * xu + e]; Kp[e] = 0.0; }
        for (int e = lane; e < nx; e += 64) { q1[e] = P.Qw[(size_t)b * nx + e] + rho; fv[e] = P.f[(size_t)b * nx + e]; }   // tiny_api.cpp:317
        for (int e = lane; e < nu; e += 64) r1[e] = P.Rw[(size_t)b * nu + e] + rho;                                        // :318
        __syncthreads();
        for (int e = lane; e < xx; e += 64) { const int i = e % nx, j = e / nx; At[j + nx * i] = A[e]; if (i == j) Pt[e] = rho; }   // :331
        for (int e = lane; e < xu; e += 64) { const int i = e % nx, j = e / nx; Bt[j + nu * i] = B[e]; }
        __syncthreads();
        int iters = 1000;
        bool fail = false;
        for (int it = 0; it < 1000; ++it) {                                   // :335
            w_mm(nu, nx, nx, Bt, Pt, BtP, lane);
            w_mm(nu, nx, nu, BtP, B, G, lane);
            for (int e = lane; e < uu; e += 64) if (e % nu == e / nu) G[e] = r1[e % nu] + G[e];   // R1 + B' P B (R1 diagonal)
            __syncthreads();
            if (!w_invert(nu, G, Gi, lu, lane)) { fail = true; break; }
            w_mm(nu, nu, nx, Gi, Bt, T1, lane);
            w_mm(nu, nx, nx, T1, Pt, T2, lane);
            w_mm(nu, nx, nx, T2, A, K, lane);                                  // Kinf           :337
            w_mm(nx, nx, nx, At, Pt, AtP, lane);
            w_mm(nx, nu, nx, B, K, BK, lane);
            for (int e = lane; e < xx; e += 64) BK[e] = A[e] - BK[e];          // A - B Kinf
            __syncthreads();
            w_mm(nx, nx, nx, AtP, BK, T3, lane);
            for (int e = lane; e < xx; e += 64) Pn[e] = ((e % nx == e / nx) ? q1[e % nx] : 0.0) + T3[e];   // Pinf :338
            double md = 0.0;
            for (int e = lane; e < xu; e += 64) md = fmax(md, fabs(K[e] - Kp[e]));
            for (int off = 32; off >= 1; off >>= 1) md = fmax(md, __shfl_xor(md, off));
            __syncthreads();
            if (md < 1e-5) { iters = it + 1; break; }                          // :340-346, break BEFORE the copies
            for (int e = lane; e < xu; e += 64) Kp[e] = K[e];
            for (int e = lane; e < xx; e += 64) Pt[e] = Pn[e];
            __syncthreads();
        }
        if (!fail) {
            w_mm(nu, nx, nx, Bt, Pn, BtP, lane);
            w_mm(nu, nx, nu, BtP, B, G, lane);
            for (int e = lane; e < uu; e += 64) if (e % nu == e / nu) G[e] = r1[e % nu] + G[e];
            __syncthreads();
            fail = !w_invert(nu, G, Gi, lu, lane);                             // Quu_inv        :352
        }
        // AmBKt = (A - B Kinf)'; APf = (AmBKt Pinf) f; BPf = (B' Pinf) f       :353-357
        w_mm(nx, nu, nx, B, K, BK, lane);
        for (int e = lane; e < xx; e += 64) { const int i = e % nx, j = e / nx; T3[j + nx * i] = A[e] - BK[e]; }
        __syncthreads();
        w_mm(nx, nx, nx, T3, Pn, AtP, lane);
        w_mm(nx, nx, 1, AtP, fv, T1, lane);                                    // APf (nx)
        w_mm(nu, nx, 1, BtP, fv, T2, lane);                                    // BPf (nu)
        for (int e = lane; e < xu; e += 64) P.Kinf[(size_t)b * xu + e] = K[e];
        for (int e = lane; e < xx; e += 64) { P.Pinf[(size_t)b * xx + e] = Pn[e]; P.AmBKt[(size_t)b * xx + e] = T3[e]; }
        for (int e = lane; e < uu; e += 64) P.Quu_inv[(size_t)b * uu + e] = Gi[e];
        for (int e = lane; e < nx; e += 64) P.APf[(size_t)b * nx + e] = T1[e];
        for (int e = lane; e < nu; e += 64) P.BPf[(size_t)b * nu + e] = T2[e];
        if (lane == 0) P.iters[b] = fail ? -1 : iters;

        // ---- lane tables of this instance (layout of admm_kernel.hip.h / tile_kernel.hip.h; see batch_tables.hip:build_tables, build_tile_tables_w)
        w_mm(nu, nu, nx, Gi, Bt, T1, lane);                                    // Quu_inv B'
        w_mm(nu, nu, 1, Gi, T2, G, lane);                                      // Quu_inv BPf  (BPf is in T2)
        const int cols = P.tab_cols, lw = P.tab_lw, tdoubles = het_tab_doubles(cols, lw);
        const int o_mb = 0, o_mf1 = cols * lw, o_mf2 = 2 * cols * lw, o_pt = 3 * cols * lw, o_vec = 4 * cols * lw;
        double* tab = P.tabs + (size_t)b * tdoubles;
        for (int e = lane; e < tdoubles; e += 64) tab[e] = 0.0;
        __syncthreads();
        for (int e = lane; e < cols * lw; e += 64) {
            const int k = e / lw, j = e % lw;              // column k, lane j
            double mb = 0.0, mf1 = 0.0, mf2 = 0.0, pt = 0.0;
            if (j < nx) {
                if (k < nx) { mb = T3[j + nx * k]; mf1 = A[j + nx * k]; pt = Pn[k + nx * j]; }
                else if (k < nx + nu) { mb = -K[(k - nx) + nu * j]; mf2 = B[j + nx * (k - nx)]; }
            } else if (j < nx + nu) {
                const int a = j - nx;
                if (k < nx) { mb = T1[a + nu * k]; mf1 = -K[a + nu * k]; }
                else if (k < nx + nu) mb = Gi[a + nu * (k - nx)];
            }
            tab[o_mb + e] = mb; tab[o_mf1 + e] = mf1; tab[o_mf2 + e] = mf2; tab[o_pt + e] = pt;
        }
        if (lane < lw) {
            const int j = lane;
            // APf was overwritten in T1 by Quu_inv B': re-read it from the output array written above
            double cb = 0.0, cf = 0.0, qr = 0.0;
            if (j < nx) { cb = P.APf[(size_t)b * nx + j]; cf = fv[j]; qr = P.Qw[(size_t)b * nx + j]; }
            else if (j < nx + nu) { cb = G[j - nx]; qr = P.Rw[(size_t)b * nu + (j - nx)]; }
            tab[o_vec + VEC_CB * lw + j] = cb; tab[o_vec + VEC_CF * lw + j] = cf;
            tab[o_vec + VEC_QR * lw + j] = qr; tab[o_vec + VEC_RHO * lw + j] = rho;
        }
        __syncthreads();
    }
}

#pragma clang fp contract(fast)
#endif  // TINYMPC_GENERAL_KERNEL_IMPL

}  // namespace tinympc_amd
