// batch_tables.hip -- the lane tables of the three kernels, built on the host from the batch's cache, bounds, cones and half-spaces:
// build_tables (one-row kernel, layout in admm_kernel.hip.h), build_tile_tables (tile kernel, TileTab<W>), build_general_tables
// (coverage kernel, GeneralArgs offsets), upload_tables.  Split off batch_dispatch.hip in round 6 (one TU held the
// tables, the helper kernels and the dispatch); declarations: batch_dispatch.hpp.
#include "batch_impl.hpp"
#include "batch_dispatch.hpp"

#include <algorithm>
#include <cstring>
#include <limits>

namespace tinympc_amd {

// Half-spaces per knot and family the register-resident LIN variants are built for: 4 (compiled in), 8 / 16 / 32
// (instantiated at run time; the tables must still fit the 64 KiB of static LDS), 0 = more than that: coverage kernel.
int lin_kmax(const TinyBatch* b) {
    int m = 0;
    if (b->set.en_state_linear) m = std::max(m, b->nsl);
    if (b->set.en_input_linear) m = std::max(m, b->nil);
    if (b->set.en_tv_state_linear) m = std::max(m, b->ntsl);
    if (b->set.en_tv_input_linear) m = std::max(m, b->ntil);
    if (m <= LIN_KMAX) return LIN_KMAX;
    // more than the compiled-in variants hold: the table stride doubles (run-time instantiated KMAX = 8, 16, 32) while
    // the tables still fit the wave's static LDS
    const bool tv = b->set.en_tv_state_linear || b->set.en_tv_input_linear;
    const bool st = b->set.en_state_linear || b->set.en_input_linear;
    for (int km = 2 * LIN_KMAX; km <= LIN_KMAX_BIG; km *= 2) {
        if (m > km) continue;
        const long lds = 8L * ((tv ? 3L * b->N * km * 16 : 1) + (st ? 3L * km * 16 : 1) + b->nx * 16 + 2L * b->N * 16);
        return lds <= 63 * 1024 ? km : 0;
    }
    return 0;
}

// Lane tables for the kernel (layout in admm_kernel.hip.h).  Rebuilt whenever cache, bounds, cones or
// the enable switches change.
void build_tables(TinyBatch* b) {
    const int nx = b->nx, nu = b->nu, N = b->N;
    const Cache& c = b->cache;
    std::vector<double>& t = b->h_tab;
    const int km = std::max(lin_kmax(b), (int)LIN_KMAX);            // stride of the half-space tables
    t.assign(tab_doubles(N, km), 0.0);
    // Quu_inv * B'  and  Quu_inv * BPf : the input rows of the fused backward step
    //   d_i = Quu_inv (B' p_{i+1} + r_i + BPf)            (admm.cpp:17)
    Mat QBt = c.Quu_inv * transpose(b->B);
    Mat QBPf = c.Quu_inv * c.BPf;
    for (int j = 0; j < nx; ++j) {                 // state lanes
        for (int k = 0; k < nx; ++k) {
            t[TAB_MB + k * 16 + j] = c.AmBKt(j, k);           // p_i += AmBKt p_{i+1}     (admm.cpp:18)
            t[TAB_MF1 + k * 16 + j] = b->A(j, k);             // x_{i+1} = A x_i ...      (admm.cpp:30)
            t[TAB_PT + k * 16 + j] = c.Pinf(k, j);            // (Xref' Pinf)[j]          (admm.cpp:292)
        }
        for (int m = 0; m < nu; ++m) {
            t[TAB_MB + (nx + m) * 16 + j] = -c.Kinf(m, j);    // - Kinf' r_i
            t[TAB_MF2 + (nx + m) * 16 + j] = b->B(j, m);      // + B u_i
        }
        t[TAB_VEC + VEC_CB * 16 + j] = c.APf(j, 0);
        t[TAB_VEC + VEC_CF * 16 + j] = b->f(j, 0);
        t[TAB_VEC + VEC_QR * 16 + j] = b->Qw[j];
        t[TAB_VEC + VEC_SMASK * 16 + j] = 1.0;
    }
    for (int a = 0; a < nu; ++a) {                 // input lanes
        const int j = nx + a;
        for (int k = 0; k < nx; ++k) {
            t[TAB_MB + k * 16 + j] = QBt(a, k);
            t[TAB_MF1 + k * 16 + j] = -c.Kinf(a, k);          // u_i = -Kinf x_i - d_i     (admm.cpp:29)
        }
        for (int m = 0; m < nu; ++m) t[TAB_MB + (nx + m) * 16 + j] = c.Quu_inv(a, m);
        t[TAB_VEC + VEC_CB * 16 + j] = QBPf(a, 0);
        t[TAB_VEC + VEC_QR * 16 + j] = b->Rw[a];
        t[TAB_VEC + VEC_NIM * 16 + j] = -1.0;
    }
    // cones (admm.cpp:102-135): lane flags
    for (int j = 0; j < 16; ++j) t[TAB_VEC + VEC_CONE_BASE * 16 + j] = -1.0;
    const bool s_on = b->set.en_state_soc && !b->Acx.empty();
    const bool i_on = b->set.en_input_soc && !b->Acu.empty();
    for (int j = 0; j < nx; ++j) t[TAB_VEC + VEC_SOCFLAG * 16 + j] = s_on ? 1.0 : 0.0;
    for (int a = 0; a < nu; ++a) t[TAB_VEC + VEC_SOCFLAG * 16 + nx + a] = i_on ? 1.0 : 0.0;
    if (b->set.en_state_soc)
        for (size_t k = 0; k < b->Acx.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) {
                t[TAB_VEC + VEC_CONE_BASE * 16 + b->Acx[k] + c3] = b->Acx[k];
                t[TAB_VEC + VEC_CONE_MU * 16 + b->Acx[k] + c3] = b->cx[k];
            }
    if (b->set.en_input_soc)
        for (size_t k = 0; k < b->Acu.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) {
                t[TAB_VEC + VEC_CONE_BASE * 16 + nx + b->Acu[k] + c3] = nx + b->Acu[k];
                t[TAB_VEC + VEC_CONE_MU * 16 + nx + b->Acu[k] + c3] = b->cu[k];
            }
    // linear-constraint slacks exist for a whole family as soon as its switch is on (admm.cpp:138-145, 176-183)
    for (int j = 0; j < nx; ++j) {
        t[TAB_VEC + VEC_LINFLAG * 16 + j] = b->set.en_state_linear ? 1.0 : 0.0;
        t[TAB_VEC + VEC_TLINFLAG * 16 + j] = b->set.en_tv_state_linear ? 1.0 : 0.0;
    }
    for (int a = 0; a < nu; ++a) {
        t[TAB_VEC + VEC_LINFLAG * 16 + nx + a] = b->set.en_input_linear ? 1.0 : 0.0;
        t[TAB_VEC + VEC_TLINFLAG * 16 + nx + a] = b->set.en_tv_input_linear ? 1.0 : 0.0;
    }
    // bounds (admm.cpp:91-98): a disabled or never-set box is (-inf, +inf)
    const double inf = std::numeric_limits<double>::infinity();
    {   // half-space tables of the LIN kernel variants: [k][16] coefficient, offset, squared norm
        auto fill = [&](double* blk, const double* Arow, int n, double bk, int lane0, bool enabled, int k) {
            double nn = 0.0;
            for (int c = 0; c < n; ++c) nn += Arow[c] * Arow[c];
            for (int c = 0; c < n; ++c) {
                blk[k * 16 + lane0 + c] = enabled ? Arow[c] : 0.0;
                blk[km * 16 + k * 16 + lane0 + c] = enabled ? bk : inf;
                blk[2 * km * 16 + k * 16 + lane0 + c] = enabled ? nn : 1.0;
            }
        };
        double* ls = &t[tab_lin_offset(N)];
        for (int k = 0; k < km; ++k)
            for (int j = 0; j < 16; ++j) { ls[km * 16 + k * 16 + j] = inf; ls[2 * km * 16 + k * 16 + j] = 1.0; }
        for (int k = 0; k < b->nsl && k < km; ++k) fill(ls, &b->Alin_x[(size_t)k * nx], nx, b->blin_x[k], 0, b->set.en_state_linear, k);
        for (int k = 0; k < b->nil && k < km; ++k) fill(ls, &b->Alin_u[(size_t)k * nu], nu, b->blin_u[k], nx, b->set.en_input_linear, k);
        double* lt = &t[tab_tlin_offset(N, km)];
        for (int s = 0; s < N; ++s) {
            double* blk = lt + (size_t)s * 3 * km * 16;
            for (int k = 0; k < km; ++k)
                for (int j = 0; j < 16; ++j) { blk[km * 16 + k * 16 + j] = inf; blk[2 * km * 16 + k * 16 + j] = 1.0; }
            for (int k = 0; k < b->ntsl && k < km; ++k)            // state lanes: slot s = knot s
                fill(blk, &b->tvA_x[((size_t)s * b->ntsl + k) * nx], nx, b->tvb_x[(size_t)s * b->ntsl + k], 0, b->set.en_tv_state_linear, k);
            if (s >= 1)                                                  // input lanes: slot s = knot s-1
                for (int k = 0; k < b->ntil && k < km; ++k)
                    fill(blk, &b->tvA_u[((size_t)(s - 1) * b->ntil + k) * nu], nu, b->tvb_u[(size_t)(s - 1) * b->ntil + k], nx, b->set.en_tv_input_linear, k);
        }
    }
    double* lo = &t[TAB_BOUNDS];
    double* hi = &t[TAB_BOUNDS + N * 16];
    for (int e = 0; e < N * 16; ++e) { lo[e] = -inf; hi[e] = inf; }
    if (b->set.en_state_bound && b->have_bounds)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < nx; ++j) {
                lo[i * 16 + j] = b->x_min[(size_t)i * nx + j];
                hi[i * 16 + j] = b->x_max[(size_t)i * nx + j];
            }
    if (b->set.en_input_bound && b->have_bounds)     // input lanes keep knot i in slot i+1 (admm_kernel.hip.h)
        for (int i = 0; i < N - 1; ++i)
            for (int a = 0; a < nu; ++a) {
                lo[(i + 1) * 16 + nx + a] = b->u_min[(size_t)i * nu + a];
                hi[(i + 1) * 16 + nx + a] = b->u_max[(size_t)i * nu + a];
            }
    // knot-invariant box? (slot 0 of the input lanes is the dummy slot and keeps (-inf, +inf): slots 1.. must agree; state lanes: 0..)
    bool uniform = N >= 2;
    for (int j = 0; j < nx + nu && uniform; ++j)
        for (int i = (j < nx ? 0 : 1); i < N && uniform; ++i)
            uniform = lo[i * 16 + j] == lo[16 + j] && hi[i * 16 + j] == hi[16 + j];
    b->bounds_uniform = uniform;
}

// Tables of the tile kernel (tile_kernel.hip.h): matrices [column k][LW = 16 W lanes], vectors [LW], bounds [N][LW]
template <int W>
static void build_tile_tables_w(TinyBatch* b) {
    using T = TileTab<W>;
    const int nx = b->nx, nu = b->nu, N = b->N, LW = T::LW;
    const Cache& c = b->cache;
    std::vector<double>& t = b->h_ttab;
    const int km = std::max(lin_kmax(b), (int)LIN_KMAX);
    t.assign(T::doubles(N, km), 0.0);
    Mat QBt = c.Quu_inv * transpose(b->B);
    Mat QBPf = c.Quu_inv * c.BPf;
    for (int j = 0; j < nx; ++j) {
        for (int k = 0; k < nx; ++k) {
            t[T::MB + k * LW + j] = c.AmBKt(j, k);
            t[T::MF1 + k * LW + j] = b->A(j, k);
            t[T::PT + k * LW + j] = c.Pinf(k, j);
        }
        for (int m = 0; m < nu; ++m) {
            t[T::MB + (nx + m) * LW + j] = -c.Kinf(m, j);
            t[T::MF2 + (nx + m) * LW + j] = b->B(j, m);
        }
        t[T::VEC + VEC_CB * LW + j] = c.APf(j, 0);
        t[T::VEC + VEC_CF * LW + j] = b->f(j, 0);
        t[T::VEC + VEC_QR * LW + j] = b->Qw[j];
        t[T::VEC + VEC_SMASK * LW + j] = 1.0;
    }
    for (int a = 0; a < nu; ++a) {
        const int j = nx + a;
        for (int k = 0; k < nx; ++k) { t[T::MB + k * LW + j] = QBt(a, k); t[T::MF1 + k * LW + j] = -c.Kinf(a, k); }
        for (int m = 0; m < nu; ++m) t[T::MB + (nx + m) * LW + j] = c.Quu_inv(a, m);
        t[T::VEC + VEC_CB * LW + j] = QBPf(a, 0);
        t[T::VEC + VEC_QR * LW + j] = b->Rw[a];
        t[T::VEC + VEC_NIM * LW + j] = -1.0;
    }
    {   // half-spaces (admm.cpp:137-211) of the LIN variants: [k][LW] coefficient, offset, squared norm (as build_tables)
        const double inf_ = std::numeric_limits<double>::infinity();
        for (int j = 0; j < nx; ++j) { t[T::VEC + VEC_LINFLAG * LW + j] = b->set.en_state_linear ? 1.0 : 0.0; t[T::VEC + VEC_TLINFLAG * LW + j] = b->set.en_tv_state_linear ? 1.0 : 0.0; }
        for (int a = 0; a < nu; ++a) { t[T::VEC + VEC_LINFLAG * LW + nx + a] = b->set.en_input_linear ? 1.0 : 0.0; t[T::VEC + VEC_TLINFLAG * LW + nx + a] = b->set.en_tv_input_linear ? 1.0 : 0.0; }
        auto fill = [&](double* blk, const double* Arow, int n, double bk, int lane0, bool enabled, int k) {
            double nn = 0.0;
            for (int c = 0; c < n; ++c) nn += Arow[c] * Arow[c];
            for (int c = 0; c < n; ++c) {
                blk[k * LW + lane0 + c] = enabled ? Arow[c] : 0.0;
                blk[km * LW + k * LW + lane0 + c] = enabled ? bk : inf_;
                blk[2 * km * LW + k * LW + lane0 + c] = enabled ? nn : 1.0;
            }
        };
        auto blank = [&](double* blk) {
            for (int k = 0; k < km; ++k)
                for (int j = 0; j < LW; ++j) { blk[km * LW + k * LW + j] = inf_; blk[2 * km * LW + k * LW + j] = 1.0; }
        };
        double* ls = &t[T::lin_offset(N)];
        blank(ls);
        for (int k = 0; k < b->nsl && k < km; ++k) fill(ls, &b->Alin_x[(size_t)k * nx], nx, b->blin_x[k], 0, b->set.en_state_linear, k);
        for (int k = 0; k < b->nil && k < km; ++k) fill(ls, &b->Alin_u[(size_t)k * nu], nu, b->blin_u[k], nx, b->set.en_input_linear, k);
        for (int sl = 0; sl < N; ++sl) {
            double* blk = &t[T::tlin_offset(N, km)] + (size_t)sl * 3 * km * LW;
            blank(blk);
            for (int k = 0; k < b->ntsl && k < km; ++k)                  // state rows: slot = knot
                fill(blk, &b->tvA_x[((size_t)sl * b->ntsl + k) * nx], nx, b->tvb_x[(size_t)sl * b->ntsl + k], 0, b->set.en_tv_state_linear, k);
            if (sl >= 1)                                                 // input rows: slot = knot + 1
                for (int k = 0; k < b->ntil && k < km; ++k)
                    fill(blk, &b->tvA_u[((size_t)(sl - 1) * b->ntil + k) * nu], nu, b->tvb_u[(size_t)(sl - 1) * b->ntil + k], nx, b->set.en_tv_input_linear, k);
        }
    }
    // cones (admm.cpp:102-135): per-row flags of the SOC variant
    for (int j = 0; j < LW; ++j) t[T::VEC + VEC_CONE_BASE * LW + j] = -1.0;
    const bool s_on = b->set.en_state_soc && !b->Acx.empty(), i_on = b->set.en_input_soc && !b->Acu.empty();
    for (int j = 0; j < nx; ++j) t[T::VEC + VEC_SOCFLAG * LW + j] = s_on ? 1.0 : 0.0;
    for (int a = 0; a < nu; ++a) t[T::VEC + VEC_SOCFLAG * LW + nx + a] = i_on ? 1.0 : 0.0;
    if (b->set.en_state_soc)
        for (size_t k = 0; k < b->Acx.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) { t[T::VEC + VEC_CONE_BASE * LW + b->Acx[k] + c3] = b->Acx[k]; t[T::VEC + VEC_CONE_MU * LW + b->Acx[k] + c3] = b->cx[k]; }
    if (b->set.en_input_soc)
        for (size_t k = 0; k < b->Acu.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) { t[T::VEC + VEC_CONE_BASE * LW + nx + b->Acu[k] + c3] = nx + b->Acu[k]; t[T::VEC + VEC_CONE_MU * LW + nx + b->Acu[k] + c3] = b->cu[k]; }
    const double inf = std::numeric_limits<double>::infinity();
    double* lo = &t[T::BOUNDS];
    double* hi = &t[T::BOUNDS + N * LW];
    for (int e = 0; e < N * LW; ++e) { lo[e] = -inf; hi[e] = inf; }
    if (b->set.en_state_bound && b->have_bounds)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < nx; ++j) { lo[i * LW + j] = b->x_min[(size_t)i * nx + j]; hi[i * LW + j] = b->x_max[(size_t)i * nx + j]; }
    if (b->set.en_input_bound && b->have_bounds)       // input lanes keep knot i in slot i+1
        for (int i = 0; i < N - 1; ++i)
            for (int a = 0; a < nu; ++a) { lo[(i + 1) * LW + nx + a] = b->u_min[(size_t)i * nu + a]; hi[(i + 1) * LW + nx + a] = b->u_max[(size_t)i * nu + a]; }
    // knot-invariant box? (as build_tables: the input lanes' slot 0 is the dummy slot) -> the UB form of the tile kernel
    bool uniform = N >= 2;
    for (int j = 0; j < nx + nu && uniform; ++j)
        for (int i = (j < nx ? 0 : 1); i < N && uniform; ++i)
            uniform = lo[i * LW + j] == lo[LW + j] && hi[i * LW + j] == hi[LW + j];
    // (one predicate for the decision -- tile_lin_variant / use_tile budget the UB form from the host bounds -- and for the launch: the
    // table is a copy of those bounds, so the two agree by construction; should they ever not, the non-UB form is the safe one)
    b->tile_bounds_uniform = uniform && box_is_uniform(b);
}

// Tables of the coverage kernel (general_kernel.hip.h): row-major [row][nz+1] matrices + vectors + constraints.
void build_general_tables(TinyBatch* b) {
    const int nx = b->nx, nu = b->nu, N = b->N, nz = nx + nu, ld = nz + 1;
    const Cache& c = b->cache;
    GeneralArgs& g = b->gargs;
    int off = 0;
    auto take = [&](int n) { int o = off; off += n; return o; };
    g.o_mb = take(nz * ld); g.o_mf1 = take(nz * ld); g.o_mf2 = take(nz * ld); g.o_pt = take(nz * ld);
    g.o_cb = take(nz); g.o_cf = take(nz); g.o_qr = take(nz);
    g.o_lo = take(N * nz); g.o_hi = take(N * nz);
    g.o_sc = take(2 * (int)b->Acx.size() + 2); g.o_ic = take(2 * (int)b->Acu.size() + 2);
    g.o_ax = take(b->nsl * nx + 1); g.o_bx = take(b->nsl + 1); g.o_au = take(b->nil * nu + 1); g.o_bu = take(b->nil + 1);
    g.o_tax = take(N * b->ntsl * nx + 1); g.o_tbx = take(N * b->ntsl + 1);
    g.o_tau = take((N - 1) * b->ntil * nu + 1); g.o_tbu = take((N - 1) * b->ntil + 1);
    std::vector<double>& t = b->h_gtab;
    t.assign(off, 0.0);
    Mat QBt = c.Quu_inv * transpose(b->B);
    Mat QBPf = c.Quu_inv * c.BPf;
    for (int j = 0; j < nx; ++j) {
        for (int k = 0; k < nx; ++k) {
            t[g.o_mb + j * ld + k] = c.AmBKt(j, k);
            t[g.o_mf1 + j * ld + k] = b->A(j, k);
            t[g.o_pt + j * ld + k] = c.Pinf(k, j);
        }
        for (int m = 0; m < nu; ++m) {
            t[g.o_mb + j * ld + nx + m] = -c.Kinf(m, j);
            t[g.o_mf2 + j * ld + nx + m] = b->B(j, m);
        }
        t[g.o_cb + j] = c.APf(j, 0); t[g.o_cf + j] = b->f(j, 0); t[g.o_qr + j] = b->Qw[j];
    }
    for (int a = 0; a < nu; ++a) {
        const int j = nx + a;
        for (int k = 0; k < nx; ++k) { t[g.o_mb + j * ld + k] = QBt(a, k); t[g.o_mf1 + j * ld + k] = -c.Kinf(a, k); }
        for (int m = 0; m < nu; ++m) t[g.o_mb + j * ld + nx + m] = c.Quu_inv(a, m);
        t[g.o_cb + j] = QBPf(a, 0); t[g.o_qr + j] = b->Rw[a];
    }
    const double inf = std::numeric_limits<double>::infinity();
    for (int e = 0; e < N * nz; ++e) { t[g.o_lo + e] = -inf; t[g.o_hi + e] = inf; }
    if (b->set.en_state_bound && b->have_bounds)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < nx; ++j) { t[g.o_lo + i * nz + j] = b->x_min[(size_t)i * nx + j]; t[g.o_hi + i * nz + j] = b->x_max[(size_t)i * nx + j]; }
    if (b->set.en_input_bound && b->have_bounds)
        for (int i = 0; i < N - 1; ++i)
            for (int a = 0; a < nu; ++a) { t[g.o_lo + i * nz + nx + a] = b->u_min[(size_t)i * nu + a]; t[g.o_hi + i * nz + nx + a] = b->u_max[(size_t)i * nu + a]; }
    for (size_t k = 0; k < b->Acx.size(); ++k) { t[g.o_sc + 2 * k] = b->Acx[k]; t[g.o_sc + 2 * k + 1] = b->cx[k]; }
    for (size_t k = 0; k < b->Acu.size(); ++k) { t[g.o_ic + 2 * k] = b->Acu[k]; t[g.o_ic + 2 * k + 1] = b->cu[k]; }
    std::copy(b->Alin_x.begin(), b->Alin_x.end(), t.begin() + g.o_ax); std::copy(b->blin_x.begin(), b->blin_x.end(), t.begin() + g.o_bx);
    std::copy(b->Alin_u.begin(), b->Alin_u.end(), t.begin() + g.o_au); std::copy(b->blin_u.begin(), b->blin_u.end(), t.begin() + g.o_bu);
    std::copy(b->tvA_x.begin(), b->tvA_x.end(), t.begin() + g.o_tax); std::copy(b->tvb_x.begin(), b->tvb_x.end(), t.begin() + g.o_tbx);
    std::copy(b->tvA_u.begin(), b->tvA_u.end(), t.begin() + g.o_tau); std::copy(b->tvb_u.begin(), b->tvb_u.end(), t.begin() + g.o_tbu);
}

int upload_tables(TinyBatch* b) {
    if (!b->tab_dirty) return TINY_OK;
    build_tables(b);
    if (b->h_tab.size() > b->d_tab_doubles) {
        HIP_TRY(b, hipStreamSynchronize(b->stream));
        (void)hipFree(b->d_tab);
        b->d_tab = nullptr; b->d_tab_doubles = 0;
        HIP_TRY(b, hipMalloc(&b->d_tab, b->h_tab.size() * sizeof(double)));
        b->d_tab_doubles = b->h_tab.size();
    }
    HIP_TRY(b, hipMemcpyAsync(b->d_tab, b->h_tab.data(), b->h_tab.size() * sizeof(double), hipMemcpyHostToDevice,
                              b->stream));
    // h_tab is pageable: the copy above is staged synchronously, so reusing h_tab later is safe
    b->tab_dirty = false;
    return TINY_OK;
}

// the tile tables of the batch's shape (W = 0, half rows, reads the one-row layout W = 1)
void build_tile_tables(TinyBatch* b) {
    if (b->tile->W <= 1) build_tile_tables_w<1>(b); else build_tile_tables_w<2>(b);
}

}  // namespace tinympc_amd
