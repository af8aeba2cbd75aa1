// adaptive_driver.cpp -- a reference-side CALLER of the adaptive-rho path, written for this repository's drop-in test.
// Compiled against the REFERENCE's headers (real Eigen types, real TinySolver structs) and linked either against the
// reference's sources (golden stdout) or against libtinympc_amd.so (the test).  It runs the quadrotor hover closed loop
// with settings->adaptive_rho = 1 and the reference's own sensitivity tables (tiny_initialize_sensitivity_matrices,
// tiny_api.hpp:54), and prints per step: iterations, solved flag, cache->rho, one entry each of the moved Kinf / Pinf /
// C1 / C2 and the applied control.
//
// Upstream reads `RhoAdapter::matrices_initialized` uninitialised (admm.cpp:339, rho_benchmark.cpp:57): scrub_stack()
// zeroes the stack region solve()'s frame is about to occupy right before every tiny_solve, which makes the reference
// take the (only non-crashing) "flag is false" path deterministically; it is a no-op for libtinympc_amd.so.
#define NSTATES 12
#define NINPUTS 4
#define NHORIZON 10

#include <cstdio>

#include <tinympc/tiny_api.hpp>

#include "problem_data/quadrotor_20hz_params.hpp"

__attribute__((noinline)) static void scrub_stack() {
    volatile char pad[1 << 16];
    for (unsigned i = 0; i < sizeof(pad); ++i) pad[i] = 0;
}

int main() {
    TinySolver* solver;
    tinyMatrix Adyn = Map<Matrix<tinytype, NSTATES, NSTATES, RowMajor>>(Adyn_data);
    tinyMatrix Bdyn = Map<Matrix<tinytype, NSTATES, NINPUTS, RowMajor>>(Bdyn_data);
    tinyVector fdyn = tinyVector::Zero(NSTATES);
    tinyVector Q = Map<Matrix<tinytype, NSTATES, 1>>(Q_data);
    tinyVector R = Map<Matrix<tinytype, NINPUTS, 1>>(R_data);
    std::cout.setstate(std::ios_base::failbit);               // "Solver converged ..." lines are not what is compared here
    if (tiny_setup(&solver, Adyn, Bdyn, fdyn, Q.asDiagonal(), R.asDiagonal(), rho_value, NSTATES, NINPUTS, NHORIZON, 0)) return 1;
    tinyMatrix x_min = tinyMatrix::Constant(NSTATES, NHORIZON, -5), x_max = tinyMatrix::Constant(NSTATES, NHORIZON, 5);
    tinyMatrix u_min = tinyMatrix::Constant(NINPUTS, NHORIZON - 1, -0.5), u_max = tinyMatrix::Constant(NINPUTS, NHORIZON - 1, 0.5);
    tiny_set_bound_constraints(solver, x_min, x_max, u_min, u_max);
    solver->settings->max_iter = 100;
    solver->settings->adaptive_rho = 1;
    solver->settings->adaptive_rho_min = 0.8;
    solver->settings->adaptive_rho_max = 50.0;
    solver->settings->adaptive_rho_enable_clipping = 1;
    tiny_initialize_sensitivity_matrices(solver);
    TinyWorkspace* work = solver->work;
    TinyCache* cache = solver->cache;
    tinyVector x0(NSTATES);
    x0 << 0, 1, 0, 0.2, 0, 0, 0.1, 0, 0, 0, 0, 0;
    tinyVector xg(NSTATES);
    xg << 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0;
    work->Xref = xg.replicate<1, NHORIZON>();
    long total = 0;
    for (int k = 0; k < 60; ++k) {
        tiny_set_x0(solver, x0);
        scrub_stack();
        tiny_solve(solver);
        total += solver->solution->iter;
        std::printf("step %2d iter %3d solved %d rho %.12e K(0,1) %.12e P(5,5) %.12e C1(0,0) %.12e C2(0,10) %.12e u0 %.9e %.9e %.9e %.9e\n", k,
                    solver->solution->iter, solver->solution->solved, cache->rho, cache->Kinf(0, 1), cache->Pinf(5, 5), cache->C1(0, 0),
                    cache->C2(0, 10), work->u(0, 0), work->u(1, 0), work->u(2, 0), work->u(3, 0));
        x0 = work->Adyn * x0 + work->Bdyn * work->u.col(0);
    }
    std::printf("total iterations %ld\n", total);
    return 0;
}
