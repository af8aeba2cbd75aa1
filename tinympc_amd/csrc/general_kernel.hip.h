// general_kernel.hip.h -- coverage kernel: the same tiny_solve() (reference src/tinympc/admm.cpp:331-455)
// for ANY (nx, nu, N) with nx + nu <= 32 and for every slack family of update_slack (box :91-98, second-order
// cone :102-135, static linear :137-173, time-varying linear :176-211).
//
// The register-resident kernel (admm_kernel.hip.h) is the fast path for the shapes it is instantiated for
// (nx+nu <= 16, N-long register arrays, box + cone).  This kernel trades speed for generality:
//   * one wavefront per instance (persistent, grid-stride), lane j = row j of the stacked knot vector for the
//     Riccati sweeps, all 64 lanes striding over the record for the element-wise phases;
//   * the slack / dual state stays in the instance's HBM records (knot-point interleaved, same layout as the
//     fast path, L2-resident across iterations); the trajectories the sequential sweeps walk (x|u, q|r, p|d)
//     are staged in LDS for the whole solve, so a sweep step never waits on global memory;
//   * the shared matrices sit in LDS ([row][nz+1] padded: conflict-free row reads), the knot vector being
//     multiplied is broadcast-read from LDS.  That formulation is LDS-bandwidth bound (see admm_kernel.hip.h)
//     -- acceptable for a coverage path, and it has no shape restriction.
// Arithmetic follows the reference's order (d_i = Quu_inv (B' p + r + BPf) uses the same pre-multiplied
// Quu_inv B' table as the fast path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tinympc_amd {

struct GeneralArgs {
    const double* gtab;       // layout below (offsets in doubles)
    const double* x0;
    const double* ref;
    double *prim, *slack, *dual, *slack_prev, *cslack, *cdual, *lslack, *ldual, *tlslack, *tldual, *qr, *pd;
    int4* status;
    double* resid;
    uint2* accum;
    double* x0_next;
    double rho, tol_pri, tol_dua;
    int batch, max_iter, check_termination;
    int nx, nu, N;
    int soc_s, soc_i;                 // cone slack active for state / input rows (en_* && num* > 0)
    int n_sc, n_ic;                   // cones projected (0 when the enable switch is off)
    int lin_s, lin_i, tlin_s, tlin_i; // enable switches (the slack exists even with 0 constraints, admm.cpp:138-145)
    int nsl, nil, ntsl, ntil;         // constraint counts (per knot for the time-varying ones)
    // offsets into gtab
    int o_mb, o_mf1, o_mf2, o_pt, o_cb, o_cf, o_qr, o_lo, o_hi, o_sc, o_ic, o_ax, o_bx, o_au, o_bu, o_tax, o_tbx,
        o_tau, o_tbu;
    // per-instance problem data (round 6): the tables tiny_batch_setup_hetero built for a register kernel -- [batch][het_stride]
    // doubles, matrices [column k][het_lw lanes] in blocks of het_cols columns (MB, MF1, MF2, PT), then the lane vectors (CB, CF, QR,
    // ... RHO) -- read TRANSPOSED into this kernel's row-major LDS matrices at the top of every instance.  null: one family (gtab)
    const double* het_tabs;
    int het_cols, het_lw, het_stride;
};

// phases of admm_phase_kernel (the batched form of the reference's exported phase functions, admm.hpp:12-17)
enum : int {
    PHASE_LINEAR_COST = 1,   // update_linear_cost     admm.cpp:262-304
    PHASE_BACKWARD = 2,      // backward_pass_grad     admm.cpp:13-20
    PHASE_FORWARD = 3,       // forward_pass           admm.cpp:25-32
    PHASE_SLACK = 4,         // update_slack           admm.cpp:81-211
    PHASE_DUAL = 5,          // update_dual            admm.cpp:219-256
    PHASE_TERMINATION = 6,   // termination_condition  admm.cpp:310-328 (the residual part; status.y = the returned bool)
};

#ifdef TINYMPC_GENERAL_KERNEL_IMPL   // the kernel body is compiled into batch_dispatch.hip only

__device__ __forceinline__ double wave_max64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// project_soc (admm.cpp:39-60) on 3 consecutive entries in place
__device__ __forceinline__ void soc3_inplace(double* s, double mu_d) {
#pragma clang fp contract(off)
    const float mu = (float)mu_d;
    const double u0 = s[2] * (double)mu;
    const double q0 = s[0] * s[0], q1 = s[1] * s[1];       // the reference is built without FMA contraction
    const float a = (float)sqrt(q0 + q1);
    if ((double)a <= -u0) { s[0] = 0.0; s[1] = 0.0; s[2] = 0.0; }
    else if ((double)a <= u0) {}
    else if ((double)a >= fabs(u0)) {
        const double scale = 0.5 * (1.0 + u0 / (double)a);
        s[0] = scale * s[0]; s[1] = scale * s[1]; s[2] = scale * (double)(a / mu);
    } else { s[0] = 0.0; s[1] = 0.0; s[2] = 0.0; }
}

// one column z (n entries, stride 1) against the half-space a'z <= b (admm.cpp:150-157, project_hyperplane :70-73)
__device__ __forceinline__ void halfspace_inplace(double* z, int n, const double* a, double b) {
    double cv = 0.0;
    for (int c = 0; c < n; ++c) cv = __dadd_rn(cv, __dmul_rn(a[c], z[c]));
    if (cv > b) {
        double nn = 0.0;
        for (int c = 0; c < n; ++c) nn = __dadd_rn(nn, __dmul_rn(a[c], a[c]));
        const double dist = (cv - b) / nn;
        for (int c = 0; c < n; ++c) z[c] = z[c] - dist * a[c];
    }
}

// LDS carve-up shared by the solve kernel and the single-phase kernel
struct GkLds {
    double *sMB, *sMF1, *sMF2, *sPT, *sW, *sU, *sPX, *sX, *sQR, *sPD;
    const double *CB, *CF, *QR;      // the lane vectors of the instance at hand (the family's, or its own: gk_load_instance)
    double rho;
};
// index of a lane vector in the register kernels' tables (admm_kernel.hip.h VEC_*; riccati_kernel.hip.h VEC_RHO)
enum : int { GK_VEC_CB = 0, GK_VEC_CF = 1, GK_VEC_QR = 2, GK_VEC_RHO = 8 };
__device__ __forceinline__ GkLds gk_carve(const GeneralArgs& P, double* lds) {
    const int nx = P.nx, nu = P.nu, N = P.N, nz = nx + nu, ld = nz + 1;
    GkLds L;
    L.sMB = lds;
    L.sMF1 = L.sMB + nz * ld;
    L.sMF2 = L.sMF1 + nz * ld;
    L.sPT = L.sMF2 + nz * ld;
    L.sW = L.sPT + nz * ld;        // [nz]  knot vector being multiplied
    L.sU = L.sW + nz;              // [nu]
    L.sPX = L.sU + nu;             // [nx]  terminal term -(Xref' Pinf)
    L.sX = L.sPX + nx;             // [N*nz] x|u trajectory (work->x, work->u)
    L.sQR = L.sX + N * nz;         // [N*nz] q|r
    L.sPD = L.sQR + N * nz;        // [N*nz] p|d
    return L;
}

// the matrices and lane vectors of instance b: the family's tables (loaded once per wave: first = true only) or, with per-instance
// problem data, the instance's own -- the register kernels' column-major lane tables read transposed.  Ends with a barrier.
__device__ __forceinline__ void gk_load_instance(const GeneralArgs& P, GkLds& L, const int b, const int lane, const bool first) {
    const int nz = P.nx + P.nu, ld = nz + 1;
    if (P.het_tabs) {
        const double* ht = P.het_tabs + (size_t)b * P.het_stride;
        const int blk = P.het_cols * P.het_lw, lw = P.het_lw;
        __syncthreads();                                     // (the instance before is done with the matrices)
        for (int e = lane; e < nz * ld; e += 64) {
            const int j = e / ld, k = e % ld;
            const bool in = k < nz;
            L.sMB[e] = in ? ht[k * lw + j] : 0.0;
            L.sMF1[e] = in ? ht[blk + k * lw + j] : 0.0;
            L.sMF2[e] = in ? ht[2 * blk + k * lw + j] : 0.0;
            L.sPT[e] = in ? ht[3 * blk + k * lw + j] : 0.0;
        }
        const double* vec = ht + 4 * blk;
        L.CB = vec + GK_VEC_CB * lw; L.CF = vec + GK_VEC_CF * lw; L.QR = vec + GK_VEC_QR * lw;
        L.rho = vec[GK_VEC_RHO * lw];
        __syncthreads();
    } else if (first) {
        for (int e = lane; e < nz * ld; e += 64) {
            L.sMB[e] = P.gtab[P.o_mb + e]; L.sMF1[e] = P.gtab[P.o_mf1 + e];
            L.sMF2[e] = P.gtab[P.o_mf2 + e]; L.sPT[e] = P.gtab[P.o_pt + e];
        }
        L.CB = P.gtab + P.o_cb; L.CF = P.gtab + P.o_cf; L.QR = P.gtab + P.o_qr;
        L.rho = P.rho;
        __syncthreads();
    }
}

// backward_pass_grad (admm.cpp:13-20) on the LDS trajectories: reads q|r and p[:,N-1], writes p and d
__device__ __forceinline__ void gk_backward(const GeneralArgs& P, const GkLds& L, const int lane) {
    const int nx = P.nx, nu = P.nu, N = P.N, nz = nx + nu, ld = nz + 1;
    const bool is_state = lane < nx, is_input = lane >= nx && lane < nz;
    const double* CB = L.CB;
    double *sMB = L.sMB, *sW = L.sW, *sQR = L.sQR, *sPD = L.sPD;
    if (is_state) sW[lane] = sPD[(N - 1) * nz + lane];
    for (int i = N - 2; i >= 0; --i) {
        const int o = i * nz;
        if (is_input) sW[lane] = sQR[o + lane];                // r_i
        __syncthreads();
        double acc = 0.0;
        if (lane < nz)
            for (int k = 0; k < nz; ++k) acc = fma(sMB[lane * ld + k], sW[k], acc);
        __syncthreads();
        if (is_state) {
            const double p = sQR[o + lane] + acc + CB[lane];   // q_i + AmBKt p - Kinf' r + APf
            sPD[o + lane] = p;
            sW[lane] = p;
        } else if (is_input) {
            sPD[o + lane] = acc + CB[lane];                    // d_i = Quu_inv (B' p + r + BPf)
        }
    }
    __syncthreads();
}

// forward_pass (admm.cpp:25-32): reads x[:,0] and d, writes u and x[:,1:]
__device__ __forceinline__ void gk_forward(const GeneralArgs& P, const GkLds& L, const int lane) {
    const int nx = P.nx, nu = P.nu, N = P.N, nz = nx + nu, ld = nz + 1;
    const bool is_state = lane < nx, is_input = lane >= nx && lane < nz;
    const double* CF = L.CF;
    double *sMF1 = L.sMF1, *sMF2 = L.sMF2, *sW = L.sW, *sU = L.sU, *sX = L.sX, *sPD = L.sPD;
    if (is_state) sW[lane] = sX[lane];
    for (int i = 0; i < N - 1; ++i) {
        const int o = i * nz;
        __syncthreads();
        double acc = 0.0;
        if (lane < nz)
            for (int k = 0; k < nx; ++k) acc = fma(sMF1[lane * ld + k], sW[k], acc);
        if (is_input) {
            const double u = acc - sPD[o + lane];              // -Kinf x_i - d_i
            sX[o + lane] = u;
            sU[lane - nx] = u;
        }
        __syncthreads();
        if (is_state) {
            double xn = acc;
            for (int m = 0; m < nu; ++m) xn = fma(sMF2[lane * ld + nx + m], sU[m], xn);
            xn += CF[lane];
            sX[o + nz + lane] = xn;
            sW[lane] = xn;
        }
    }
    __syncthreads();
}

// cone and half-space projections of the refreshed slacks (admm.cpp:112-211): one lane per knot point,
// constraints applied sequentially
__device__ __forceinline__ void gk_project(const GeneralArgs& P, const size_t rec, const int lane) {
    const int nx = P.nx, nu = P.nu, N = P.N, nz = nx + nu;
    for (int i = lane; i < N; i += 64) {
        double* vcol;
        double zbuf[32];
        const size_t o = rec + (size_t)i * nz;
        for (int k = 0; k < P.n_sc; ++k) {
            vcol = P.cslack + o + (int)P.gtab[P.o_sc + 2 * k];
            double s3[3] = {vcol[0], vcol[1], vcol[2]};
            soc3_inplace(s3, P.gtab[P.o_sc + 2 * k + 1]);
            vcol[0] = s3[0]; vcol[1] = s3[1]; vcol[2] = s3[2];
        }
        if (i < N - 1)
            for (int k = 0; k < P.n_ic; ++k) {
                vcol = P.cslack + o + nx + (int)P.gtab[P.o_ic + 2 * k];
                double s3[3] = {vcol[0], vcol[1], vcol[2]};
                soc3_inplace(s3, P.gtab[P.o_ic + 2 * k + 1]);
                vcol[0] = s3[0]; vcol[1] = s3[1]; vcol[2] = s3[2];
            }
        if (P.lin_s && P.nsl > 0) {
            for (int c = 0; c < nx; ++c) zbuf[c] = P.lslack[o + c];
            for (int k = 0; k < P.nsl; ++k) halfspace_inplace(zbuf, nx, P.gtab + P.o_ax + k * nx, P.gtab[P.o_bx + k]);
            for (int c = 0; c < nx; ++c) P.lslack[o + c] = zbuf[c];
        }
        if (P.lin_i && P.nil > 0 && i < N - 1) {
            for (int c = 0; c < nu; ++c) zbuf[c] = P.lslack[o + nx + c];
            for (int k = 0; k < P.nil; ++k) halfspace_inplace(zbuf, nu, P.gtab + P.o_au + k * nu, P.gtab[P.o_bu + k]);
            for (int c = 0; c < nu; ++c) P.lslack[o + nx + c] = zbuf[c];
        }
        if (P.tlin_s && P.ntsl > 0) {
            for (int c = 0; c < nx; ++c) zbuf[c] = P.tlslack[o + c];
            for (int k = 0; k < P.ntsl; ++k)
                halfspace_inplace(zbuf, nx, P.gtab + P.o_tax + (size_t)(i * P.ntsl + k) * nx, P.gtab[P.o_tbx + i * P.ntsl + k]);
            for (int c = 0; c < nx; ++c) P.tlslack[o + c] = zbuf[c];
        }
        if (P.tlin_i && P.ntil > 0 && i < N - 1) {
            for (int c = 0; c < nu; ++c) zbuf[c] = P.tlslack[o + nx + c];
            for (int k = 0; k < P.ntil; ++k)
                halfspace_inplace(zbuf, nu, P.gtab + P.o_tau + (size_t)(i * P.ntil + k) * nu, P.gtab[P.o_tbu + i * P.ntil + k]);
            for (int c = 0; c < nu; ++c) P.tlslack[o + nx + c] = zbuf[c];
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void admm_general_kernel(const GeneralArgs P) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int nx = P.nx, nu = P.nu, N = P.N, nz = nx + nu, ld = nz + 1;
    GkLds L = gk_carve(P, lds);
    double *sPT = L.sPT, *sPX = L.sPX, *sX = L.sX, *sQR = L.sQR, *sPD = L.sPD;
    const double* LO = P.gtab + P.o_lo;
    const double* HI = P.gtab + P.o_hi;
    const bool is_state = lane < nx;
    const int rec_n = N * nz;

    for (int b = blockIdx.x; b < P.batch; b += gridDim.x) {
        gk_load_instance(P, L, b, lane, b == (int)blockIdx.x);
        const double* QR = L.QR;
        const double rho = L.rho;
        const size_t rec = (size_t)b * rec_n;
        // ---- per-solve setup: x[:,0] = x0, terminal term, slack initialisation (admm.cpp:352-376)
        for (int e = lane; e < rec_n; e += 64) sX[e] = P.prim[rec + e];   // previous solve's x|u (cone / linear slack init)
        __syncthreads();
        if (is_state) {
            const double x0v = P.x0[(size_t)b * nx + lane];
            sX[lane] = x0v;
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc = fma(P.ref[rec + (size_t)(N - 1) * nz + k], sPT[lane * ld + k], acc);
            sPX[lane] = -acc;                                          // admm.cpp:292
        }
        __syncthreads();
        for (int e = lane; e < rec_n; e += 64) {
            const int i = e / nz, j = e - i * nz;
            const bool st = j < nx;
            if (!st && i == N - 1) continue;
            const double xv = sX[e];
            if (st ? P.soc_s : P.soc_i) P.cslack[rec + e] = xv;        // vcnew = x / zcnew = u
            if (st ? P.lin_s : P.lin_i) P.lslack[rec + e] = xv;        // vlnew = x / zlnew = u
            if (st ? P.tlin_s : P.tlin_i) P.tlslack[rec + e] = xv;     // vlnew_tv / zlnew_tv
        }
        __syncthreads();

        int iter = 0, solved = 0, checked = 0, countdown = P.check_termination;
        double r_ps = 0.0, r_pi = 0.0, r_ds = 0.0, r_di = 0.0;
        for (int it = 0; it < P.max_iter; ++it) {
            // ---- v = vnew of the previous iteration (admm.cpp:445-446), then update_linear_cost (:262-302)
            for (int e = lane; e < rec_n; e += 64) {
                const int i = e / nz, j = e - i * nz;
                const bool st = j < nx;
                if (!st && i == N - 1) continue;
                const double vn = P.slack[rec + e];
                if (it > 0) P.slack_prev[rec + e] = vn;
                double qv = -(P.ref[rec + e] * QR[j]);                                            // :266 / :279
                qv -= rho * (vn - P.dual[rec + e]);                                               // :267 / :280
                double pv = (st && i == N - 1) ? (sPX[j] - rho * (vn - P.dual[rec + e])) : 0.0;   // :292-293
                if (st ? P.soc_s : P.soc_i) {
                    const double tt = rho * (P.cslack[rec + e] - P.cdual[rec + e]);               // :269 / :282 / :295
                    qv -= tt; pv -= tt;
                }
                if (st ? P.lin_s : P.lin_i) {
                    const double tt = rho * (P.lslack[rec + e] - P.ldual[rec + e]);               // :272 / :285 / :298
                    qv -= tt; pv -= tt;
                }
                if (st ? P.tlin_s : P.tlin_i) {
                    const double tt = rho * (P.tlslack[rec + e] - P.tldual[rec + e]);             // :275 / :288 / :301
                    qv -= tt; pv -= tt;
                }
                sQR[e] = qv;
                if (st && i == N - 1) sPD[e] = pv;
            }
            __syncthreads();
            // ---- backward_pass_grad (admm.cpp:13-20)
            gk_backward(P, L, lane);
            // ---- forward_pass (admm.cpp:25-32)
            gk_forward(P, L, lane);
            // ---- update_slack (box) + update_dual + residuals (admm.cpp:85-98, 222-225, 314-317)
            double m_ps = 0.0, m_pi = 0.0, m_ds = 0.0, m_di = 0.0;
            for (int e = lane; e < rec_n; e += 64) {
                const int i = e / nz, j = e - i * nz;
                const bool st = j < nx;
                if (!st && i == N - 1) continue;
                const double xv = sX[e];
                const double t = xv + P.dual[rec + e];
                const double vn = fmin(HI[i * nz + j], fmax(LO[i * nz + j], t));
                const double pr = fabs(xv - vn), du = fabs(P.slack_prev[rec + e] - vn);
                if (st) { m_ps = fmax(m_ps, pr); m_ds = fmax(m_ds, du); }
                else { m_pi = fmax(m_pi, pr); m_di = fmax(m_di, du); }
                P.dual[rec + e] = t - vn;
                P.slack[rec + e] = vn;
                // cone / linear slacks: refresh (admm.cpp:102-109, 138-145, 176-183); projected below
                if (st ? P.soc_s : P.soc_i) P.cslack[rec + e] = xv + P.cdual[rec + e];
                if (st ? P.lin_s : P.lin_i) P.lslack[rec + e] = xv + P.ldual[rec + e];
                if (st ? P.tlin_s : P.tlin_i) P.tlslack[rec + e] = xv + P.tldual[rec + e];
            }
            __syncthreads();
            // ---- projections (admm.cpp:112-211)
            gk_project(P, rec, lane);
            // ---- duals of the cone / linear slacks (admm.cpp:228-255)
            if (P.soc_s | P.soc_i | P.lin_s | P.lin_i | P.tlin_s | P.tlin_i) {
                for (int e = lane; e < rec_n; e += 64) {
                    const int i = e / nz, j = e - i * nz;
                    const bool st = j < nx;
                    if (!st && i == N - 1) continue;
                    const double xv = sX[e];
                    if (st ? P.soc_s : P.soc_i) P.cdual[rec + e] = (P.cdual[rec + e] + xv) - P.cslack[rec + e];
                    if (st ? P.lin_s : P.lin_i) P.ldual[rec + e] = (P.ldual[rec + e] + xv) - P.lslack[rec + e];
                    if (st ? P.tlin_s : P.tlin_i) P.tldual[rec + e] = (P.tldual[rec + e] + xv) - P.tlslack[rec + e];
                }
                __syncthreads();
            }
            iter += 1;
            // ---- termination_condition (admm.cpp:310-328): box slack only
            bool conv = false;
            if (countdown > 0 && --countdown == 0) {
                countdown = P.check_termination;
                checked = 1;
                r_ps = wave_max64(m_ps); r_pi = wave_max64(m_pi);
                r_ds = wave_max64(m_ds) * rho; r_di = wave_max64(m_di) * rho;
                conv = (r_ps < P.tol_pri) && (r_pi < P.tol_pri) && (r_ds < P.tol_dua) && (r_di < P.tol_dua);
            }
            if (conv) { solved = 1; break; }
        }
        if (!solved && iter > 0) {                                     // the last v = vnew (admm.cpp:445-446)
            for (int e = lane; e < rec_n; e += 64) {
                const int i = e / nz, j = e - i * nz;
                if (j >= nx && i == N - 1) continue;
                P.slack_prev[rec + e] = P.slack[rec + e];
            }
        }
        for (int e = lane; e < rec_n; e += 64) {                       // trajectories back to the HBM records
            const int i = e / nz, j = e - i * nz;
            if (j >= nx && i == N - 1) continue;
            P.prim[rec + e] = sX[e];
            if (iter > 0) {                                            // max_iter = 0: q, r, p, d were never computed
                P.qr[rec + e] = sQR[e];
                P.pd[rec + e] = sPD[e];
            }
        }
        if (P.x0_next && iter > 0 && is_state) P.x0_next[(size_t)b * nx + lane] = sX[nz + lane];
        if (lane == 0) {
            P.status[b] = make_int4(iter, solved, solved ? 1 : 11, checked);
            *reinterpret_cast<double4*>(P.resid + (size_t)b * 4) = make_double4(r_ps, r_pi, r_ds, r_di);
            if (P.accum) {
                uint2 ac = P.accum[b];
                ac.x += (unsigned)iter;
                ac.y += (unsigned)solved;
                P.accum[b] = ac;
            }
        }
        __syncthreads();
    }
}


// ---- one phase of the iteration over the batch: the batched form of the phase functions the reference exports
// (admm.hpp:12-17).  Works on the HBM records only (x|u = prim, q|r, p|d, the slack / dual families), exactly the
// workspace fields the reference function reads and writes; one wavefront per instance.

__global__ __launch_bounds__(64) void admm_phase_kernel(const GeneralArgs P, const int phase) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int nx = P.nx, nu = P.nu, N = P.N, nz = nx + nu, ld = nz + 1;
    GkLds L = gk_carve(P, lds);
    const double* LO = P.gtab + P.o_lo;
    const double* HI = P.gtab + P.o_hi;
    const int rec_n = N * nz;
    for (int b = blockIdx.x; b < P.batch; b += gridDim.x) {
        gk_load_instance(P, L, b, lane, b == (int)blockIdx.x);
        const double* QR = L.QR;
        const double rho = L.rho;
        const size_t rec = (size_t)b * rec_n;
        for (int e = lane; e < rec_n; e += 64) { L.sX[e] = P.prim[rec + e]; L.sQR[e] = P.qr[rec + e]; L.sPD[e] = P.pd[rec + e]; }
        __syncthreads();
        if (phase == PHASE_LINEAR_COST) {
            if (lane < nx) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc = fma(P.ref[rec + (size_t)(N - 1) * nz + k], L.sPT[lane * ld + k], acc);
                L.sPX[lane] = -acc;                                                               // :292
            }
            __syncthreads();
            for (int e = lane; e < rec_n; e += 64) {
                const int i = e / nz, j = e - i * nz;
                const bool st = j < nx;
                if (!st && i == N - 1) continue;
                const double vn = P.slack[rec + e];
                double qv = -(P.ref[rec + e] * QR[j]);                                            // :266 / :279
                qv -= rho * (vn - P.dual[rec + e]);                                               // :267 / :280
                double pv = (st && i == N - 1) ? (L.sPX[j] - rho * (vn - P.dual[rec + e])) : 0.0; // :292-293
                if (st ? P.soc_s : P.soc_i) { const double tt = rho * (P.cslack[rec + e] - P.cdual[rec + e]); qv -= tt; pv -= tt; }
                if (st ? P.lin_s : P.lin_i) { const double tt = rho * (P.lslack[rec + e] - P.ldual[rec + e]); qv -= tt; pv -= tt; }
                if (st ? P.tlin_s : P.tlin_i) { const double tt = rho * (P.tlslack[rec + e] - P.tldual[rec + e]); qv -= tt; pv -= tt; }
                L.sQR[e] = qv;
                if (st && i == N - 1) L.sPD[e] = pv;
            }
            __syncthreads();
        } else if (phase == PHASE_BACKWARD) {
            gk_backward(P, L, lane);
        } else if (phase == PHASE_FORWARD) {
            gk_forward(P, L, lane);
        } else if (phase == PHASE_SLACK) {
            for (int e = lane; e < rec_n; e += 64) {
                const int i = e / nz, j = e - i * nz;
                const bool st = j < nx;
                if (!st && i == N - 1) continue;
                const double xv = L.sX[e];
                const double t = xv + P.dual[rec + e];                                            // :85 / :88
                P.slack[rec + e] = fmin(HI[i * nz + j], fmax(LO[i * nz + j], t));                 // :91-98 (+-inf when disabled)
                if (st ? P.soc_s : P.soc_i) P.cslack[rec + e] = xv + P.cdual[rec + e];            // :102-109
                if (st ? P.lin_s : P.lin_i) P.lslack[rec + e] = xv + P.ldual[rec + e];            // :138-145
                if (st ? P.tlin_s : P.tlin_i) P.tlslack[rec + e] = xv + P.tldual[rec + e];        // :176-183
            }
            __syncthreads();
            gk_project(P, rec, lane);
        } else if (phase == PHASE_DUAL) {
            for (int e = lane; e < rec_n; e += 64) {
                const int i = e / nz, j = e - i * nz;
                const bool st = j < nx;
                if (!st && i == N - 1) continue;
                const double xv = L.sX[e];
                P.dual[rec + e] = (P.dual[rec + e] + xv) - P.slack[rec + e];                      // :222 / :225
                if (st ? P.soc_s : P.soc_i) P.cdual[rec + e] = (P.cdual[rec + e] + xv) - P.cslack[rec + e];
                if (st ? P.lin_s : P.lin_i) P.ldual[rec + e] = (P.ldual[rec + e] + xv) - P.lslack[rec + e];
                if (st ? P.tlin_s : P.tlin_i) P.tldual[rec + e] = (P.tldual[rec + e] + xv) - P.tlslack[rec + e];
            }
            __syncthreads();
        } else if (phase == PHASE_TERMINATION) {
            double m_ps = 0.0, m_pi = 0.0, m_ds = 0.0, m_di = 0.0;
            for (int e = lane; e < rec_n; e += 64) {
                const int i = e / nz, j = e - i * nz;
                const bool st = j < nx;
                if (!st && i == N - 1) continue;
                const double vn = P.slack[rec + e];
                const double pr = fabs(L.sX[e] - vn), du = fabs(P.slack_prev[rec + e] - vn);      // :314-317
                if (st) { m_ps = fmax(m_ps, pr); m_ds = fmax(m_ds, du); }
                else { m_pi = fmax(m_pi, pr); m_di = fmax(m_di, du); }
            }
            const double r_ps = wave_max64(m_ps), r_pi = wave_max64(m_pi);
            const double r_ds = wave_max64(m_ds) * rho, r_di = wave_max64(m_di) * rho;
            const bool conv = (r_ps < P.tol_pri) && (r_pi < P.tol_pri) && (r_ds < P.tol_dua) && (r_di < P.tol_dua);
            if (lane == 0) {
                int4 st4 = P.status[b];
                st4.y = conv ? 1 : 0; st4.w = 1;
                P.status[b] = st4;
                *reinterpret_cast<double4*>(P.resid + (size_t)b * 4) = make_double4(r_ps, r_pi, r_ds, r_di);
            }
        }
        for (int e = lane; e < rec_n; e += 64) {
            const int i = e / nz, j = e - i * nz;
            if (j >= nx && i == N - 1) continue;
            P.prim[rec + e] = L.sX[e]; P.qr[rec + e] = L.sQR[e]; P.pd[rec + e] = L.sPD[e];
        }
        __syncthreads();
    }
}

// project_soc (admm.cpp:39-60) / project_hyperplane (:70-73) on one small vector (the exported utility functions)
__global__ void project_soc_kernel(double* s, const int n, const float mu) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || n < 1) return;
    const double u0 = s[n - 1] * (double)mu;                                            // :40
    double nn = 0.0;
    for (int c = 0; c < n - 1; ++c) { const double q = s[c] * s[c]; nn = nn + q; }
    const float a = (float)sqrt(nn);                                                    // :42
    if ((double)a <= -u0) { for (int c = 0; c < n; ++c) s[c] = 0.0; }                   // :46
    else if ((double)a <= u0) {}                                                        // :49
    else if ((double)a >= fabs(u0)) {                                                   // :52
        const double scale = 0.5 * (1.0 + u0 / (double)a);
        for (int c = 0; c < n - 1; ++c) s[c] = scale * s[c];
        s[n - 1] = scale * (double)(a / mu);
    } else { for (int c = 0; c < n; ++c) s[c] = 0.0; }
}
__global__ void project_hyperplane_kernel(double* z, const double* a, const int n, const double b) {
    if (threadIdx.x != 0) return;
    double az = 0.0, aa = 0.0;
    for (int c = 0; c < n; ++c) { az = __dadd_rn(az, __dmul_rn(a[c], z[c])); aa = __dadd_rn(aa, __dmul_rn(a[c], a[c])); }
    const double dist = (az - b) / aa;                                                  // :71
    for (int c = 0; c < n; ++c) z[c] = z[c] - dist * a[c];                              // :72
}

#endif  // TINYMPC_GENERAL_KERNEL_IMPL

}  // namespace tinympc_amd
