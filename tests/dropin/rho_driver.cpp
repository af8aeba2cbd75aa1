// rho_driver.cpp -- a reference-side CALLER of the helpers rho_benchmark.hpp declares (initialize_format_matrices,
// format_matrices, compute_residuals, predict_rho, update_matrices_with_derivatives, benchmark_rho_adaptation, micros),
// written for this repository's drop-in test.  Compiled against the REFERENCE's headers (real Eigen types, the real
// RhoAdapter) and linked either against the reference's sources (golden stdout) or against libtinympc_amd.so (the test):
// the C++-mangled symbols must resolve and print the same numbers.
#define NSTATES 12
#define NINPUTS 4
#define NHORIZON 10

#include <cstdio>

#include <tinympc/tiny_api.hpp>
#include <tinympc/rho_benchmark.hpp>

#include "problem_data/quadrotor_20hz_params.hpp"

uint32_t micros();                                        // defined by rho_benchmark.cpp:9-11 (no header declares it)

static unsigned long long lcg = 88172645463325252ull;
static double draw() {                                     // xorshift64: the same stream whichever library is linked
    lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17;
    return (double)(lcg >> 11) / 9007199254740992.0 - 0.5;
}
static void fill(tinyMatrix& m, double scale) {
    for (int c = 0; c < m.cols(); ++c)
        for (int r = 0; r < m.rows(); ++r) m(r, c) = scale * draw();
}
static double absmax(const tinyMatrix& m) { return m.cwiseAbs().maxCoeff(); }

int main() {
    TinySolver* solver;
    tinyMatrix Adyn = Map<Matrix<tinytype, NSTATES, NSTATES, RowMajor>>(Adyn_data);
    tinyMatrix Bdyn = Map<Matrix<tinytype, NSTATES, NINPUTS, RowMajor>>(Bdyn_data);
    tinyVector fdyn = tinyVector::Zero(NSTATES);
    tinyVector Q = Map<Matrix<tinytype, NSTATES, 1>>(Q_data);
    tinyVector R = Map<Matrix<tinytype, NINPUTS, 1>>(R_data);
    if (tiny_setup(&solver, Adyn, Bdyn, fdyn, Q.asDiagonal(), R.asDiagonal(), rho_value, NSTATES, NINPUTS, NHORIZON, 0)) return 1;
    tiny_initialize_sensitivity_matrices(solver);
    TinyWorkspace* work = solver->work;
    TinyCache* cache = solver->cache;

    std::printf("micros %u\n", (unsigned)micros());
    RhoAdapter adapter;
    adapter.rho_min = 4.9; adapter.rho_max = 40.0; adapter.clip = true; adapter.matrices_initialized = false;
    initialize_format_matrices(&adapter, NSTATES, NINPUTS, NHORIZON);
    std::printf("init: A %ldx%ld P %ldx%ld z %ld x %ld dims %d %d %d flag %d zero %d\n", (long)adapter.A_matrix.rows(), (long)adapter.A_matrix.cols(),
                (long)adapter.P_matrix.rows(), (long)adapter.P_matrix.cols(), (long)adapter.z_vector.rows(), (long)adapter.x_decision.rows(),
                adapter.format_nx, adapter.format_nu, adapter.format_N, (int)adapter.matrices_initialized,
                (int)(absmax(adapter.A_matrix) == 0.0 && absmax(adapter.ATy_vector) == 0.0));

    tinyMatrix x(NSTATES, NHORIZON), u(NINPUTS, NHORIZON - 1), v(NSTATES, NHORIZON), z(NINPUTS, NHORIZON - 1), g(NSTATES, NHORIZON), y(NINPUTS, NHORIZON - 1);
    for (int round = 0; round < 3; ++round) {
        fill(x, 1.0); fill(u, 0.6); fill(v, 1.0); fill(z, 0.6); fill(g, 0.1 * (round + 1)); fill(y, 0.1 * (round + 1));
        format_matrices(&adapter, x, u, v, z, g, y, cache, work, NHORIZON);
        std::printf("format %d: |A| %.12e sumA %.12e |P| %.12e traceP %.12e |q| %.12e |z| %.12e |y| %.12e x[17] %.12e\n", round, absmax(adapter.A_matrix),
                    adapter.A_matrix.sum(), absmax(adapter.P_matrix), adapter.P_matrix.trace(), absmax(adapter.q_vector), absmax(adapter.z_vector),
                    absmax(adapter.y_vector), adapter.x_decision(17, 0));
        tinytype pri_res, dual_res, pri_norm, dual_norm;
        compute_residuals(&adapter, &pri_res, &dual_res, &pri_norm, &dual_norm);
        std::printf("residuals %d: %.10e %.10e %.10e %.10e | Ax[5] %.10e r_prim[40] %.10e Px[150] %.10e ATy[3] %.10e r_dual[77] %.10e\n", round, pri_res,
                    dual_res, pri_norm, dual_norm, adapter.Ax_vector(5, 0), adapter.r_prim_vector(40, 0), adapter.Px_vector(150, 0), adapter.ATy_vector(3, 0),
                    adapter.r_dual_vector(77, 0));
        const tinytype nr = predict_rho(&adapter, pri_res, dual_res, pri_norm, dual_norm, cache->rho);
        adapter.clip = false;
        const tinytype nr_free = predict_rho(&adapter, pri_res, dual_res, pri_norm, dual_norm, cache->rho);
        adapter.clip = true;
        std::printf("predict %d: %.12e (clipped) %.12e (free)\n", round, nr, nr_free);
        update_matrices_with_derivatives(cache, nr);
        std::printf("update %d: rho %.12e K(0,1) %.12e P(5,5) %.12e C1(0,0) %.12e C2(0,10) %.12e\n", round, cache->rho, cache->Kinf(0, 1), cache->Pinf(5, 5),
                    cache->C1(0, 0), cache->C2(0, 10));
    }
    // the whole step in one call, on an adapter whose matrices do not exist yet (format_matrices creates them)
    RhoAdapter fresh;
    fresh.rho_min = 1.0; fresh.rho_max = 100.0; fresh.clip = true; fresh.matrices_initialized = false;
    RhoBenchmarkResult res;
    fill(x, 1.0); fill(u, 0.6); fill(v, 1.0); fill(z, 0.6); fill(g, 0.2); fill(y, 0.2);
    benchmark_rho_adaptation(&fresh, x, u, v, z, g, y, cache, work, NHORIZON, &res);
    std::printf("benchmark: time %u initial %.12e final %.12e pri %.10e dual %.10e pnorm %.10e dnorm %.10e | rho %.12e K(3,7) %.12e P(0,0) %.12e flag %d\n",
                (unsigned)res.time_us, res.initial_rho, res.final_rho, res.pri_res, res.dual_res, res.pri_norm, res.dual_norm, cache->rho, cache->Kinf(3, 7),
                cache->Pinf(0, 0), (int)fresh.matrices_initialized);
    return 0;
}
