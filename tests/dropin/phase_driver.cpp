// phase_driver.cpp -- a reference-side CALLER of the phase functions the reference exports (admm.hpp:12-34),
// written for this repository's drop-in test.  It is compiled against the REFERENCE's headers (real Eigen types,
// real TinySolver structs) and linked either against the reference's sources (golden stdout) or against
// libtinympc_amd.so (the test): every Eigen by-value argument / return value crosses the boundary exactly as in a
// user's program.  Prints every workspace field after every phase with 12 significant digits.
#define NSTATES 6
#define NINPUTS 3
#define NHORIZON 10

#include <cstdio>
#include <cmath>

#include <tinympc/tiny_api.hpp>
#include <tinympc/admm.hpp>

#include "problem_data/rocket_landing_params_20hz.hpp"

static unsigned long long lcg = 0x2545F4914F6CDD1DULL;
static double rnd() {                       // deterministic, identical in both builds
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)(lcg >> 11) / 9007199254740992.0 - 0.5) * 2.0;
}
static void fill(tinyMatrix& m, double scale) {
    for (int j = 0; j < m.cols(); ++j)
        for (int i = 0; i < m.rows(); ++i) m(i, j) = scale * rnd();
}
static void show(const char* phase, const char* name, const tinyMatrix& m) {
    std::printf("%s %s", phase, name);
    for (int j = 0; j < m.cols(); ++j)
        for (int i = 0; i < m.rows(); ++i) std::printf(" %.12e", m(i, j));
    std::printf("\n");
}
static void dump(const char* phase, TinyWorkspace* w) {
    show(phase, "x", w->x); show(phase, "u", w->u); show(phase, "q", w->q); show(phase, "r", w->r);
    show(phase, "p", w->p); show(phase, "d", w->d); show(phase, "vnew", w->vnew); show(phase, "znew", w->znew);
    show(phase, "g", w->g); show(phase, "y", w->y); show(phase, "vcnew", w->vcnew); show(phase, "zcnew", w->zcnew);
    show(phase, "gc", w->gc); show(phase, "yc", w->yc);
}

int main() {
    const int nx = 6, nu = 3, N = 10;
    TinySolver* solver;
    tinyMatrix Adyn = Eigen::Map<Eigen::Matrix<tinytype, 6, 6, Eigen::RowMajor>>(Adyn_data);
    tinyMatrix Bdyn = Eigen::Map<Eigen::Matrix<tinytype, 6, 3, Eigen::RowMajor>>(Bdyn_data);
    tinyVector fdyn = Eigen::Map<Eigen::Matrix<tinytype, 6, 1>>(fdyn_data);
    tinyVector Q = Eigen::Map<Eigen::Matrix<tinytype, 6, 1>>(Q_data);
    tinyVector R = Eigen::Map<Eigen::Matrix<tinytype, 3, 1>>(R_data);
    if (tiny_setup(&solver, Adyn, Bdyn, fdyn, Q.asDiagonal(), R.asDiagonal(), rho_value, nx, nu, N, 0)) return 1;
    tinyMatrix x_min = tinyMatrix::Constant(nx, N, -0.6), x_max = tinyMatrix::Constant(nx, N, 0.7);
    tinyMatrix u_min = tinyMatrix::Constant(nu, N - 1, -0.3), u_max = tinyMatrix::Constant(nu, N - 1, 0.4);
    tiny_set_bound_constraints(solver, x_min, x_max, u_min, u_max);
    tinyVector cx(1), cu(1);
    cx << 0.7; cu << 0.4;
    VectorXi Acx(1), qcx(1), Acu(1), qcu(1);
    Acx << 1; qcx << 3; Acu << 0; qcu << 3;
    tiny_set_cone_constraints(solver, Acx, qcx, cx, Acu, qcu, cu);       // definition order: state triple first
    solver->settings->en_state_soc = 1;
    solver->settings->en_input_soc = 1;
    TinyWorkspace* w = solver->work;
    fill(w->Xref, 0.5); fill(w->Uref, 0.5); fill(w->x, 0.5); fill(w->u, 0.5); fill(w->q, 0.5); fill(w->r, 0.5);
    fill(w->p, 0.5); fill(w->d, 0.5); fill(w->v, 0.5); fill(w->vnew, 0.5); fill(w->z, 0.5); fill(w->znew, 0.5);
    fill(w->g, 0.5); fill(w->y, 0.5); fill(w->vcnew, 0.5); fill(w->zcnew, 0.5); fill(w->gc, 0.5); fill(w->yc, 0.5);

    update_linear_cost(solver); dump("update_linear_cost", w);
    backward_pass_grad(solver); dump("backward_pass_grad", w);
    forward_pass(solver); dump("forward_pass", w);
    update_slack(solver); dump("update_slack", w);
    update_dual(solver); dump("update_dual", w);
    w->iter = 4;
    solver->settings->check_termination = 2;
    const bool t = termination_condition(solver);
    std::printf("termination %d %.12e %.12e %.12e %.12e\n", (int)t, w->primal_residual_state, w->dual_residual_state,
                w->primal_residual_input, w->dual_residual_input);
    w->iter = 5;
    w->primal_residual_state = -1.0;
    std::printf("termination_skipped %d %.3f\n", (int)termination_condition(solver), w->primal_residual_state);

    // the projection utilities: Eigen vectors by value / by reference in, Eigen vector by value out
    const double cases[5][3] = {{0.3, 0.4, 10.0}, {0.3, 0.4, -10.0}, {3.0, 4.0, 1.0}, {0.0, 0.0, 0.0}, {-1.5, 0.25, 0.75}};
    for (int c = 0; c < 5; ++c) {
        tinyVector s(3);
        s << cases[c][0], cases[c][1], cases[c][2];
        tinyVector o = project_soc(s, 0.5f);
        std::printf("project_soc %d %.12e %.12e %.12e (input kept: %.3f)\n", c, o(0), o(1), o(2), s(2));
    }
    tinyVector z(4), a(4);
    z << 1.0, -2.0, 0.5, 3.0;
    a << 0.5, 0.25, -1.0, 2.0;
    tinyVector h = project_hyperplane(z, a, 1.5);
    std::printf("project_hyperplane %.12e %.12e %.12e %.12e residual %.3e\n", h(0), h(1), h(2), h(3), std::fabs(a.dot(h) - 1.5));
    return 0;
}
