// ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A flat extern-"C" facade over the REAL TinyMPC reference (compiled from the
// sources where they lie under /root/reference by oracle/Makefile; nothing from
// the reference is copied into this repository).  It mirrors the entry points of
// oracle/tinympc_oracle.h one-for-one (prefix ref_ instead of oracle_) so the same
// Python driver (oracle/cpu_solvers.py) can run either implementation:
//   * oracle/gen_golden.py uses it to generate tests/golden/*.npz,
//   * tests/test_oracle_vs_ref.py uses it (when built) to validate the C restatement,
//   * bench.py uses it (when built) as cpu_baseline kind "reference".
// The resulting oracle/_ref/libtinympc_ref.so is git-ignored but travels to the GPU
// box with the repo snapshot (Eigen is header-only, so it has no run-time
// dependency on /root/reference).
#include <cstring>
#include <iostream>
#include <string>

#include <tinympc/tiny_api.hpp>   // reference header, found via -I/root/reference/src

namespace {
struct RefHandle {
    TinySolver* solver;
};

inline Eigen::Map<const tinyMatrix> cm(const double* p, int r, int c) {
    return Eigen::Map<const tinyMatrix>(p, r, c);
}
}  // namespace

extern "C" {

// The reference prints one line per converged solve from inside the hot loop
// (src/tinympc/admm.cpp:439) and the setup prints when verbose; mute std::cout so
// timing the CPU baseline measures arithmetic, not stdio.  The stream state is process-global in a shared libstdc++, and the
// product library prints its own "Solver converged" lines through std::cout: oracle/Makefile therefore links THIS library
// against a private static libstdc++ (-static-libstdc++ -Wl,--exclude-libs,ALL), so the failbit set here is on the
// reference's own std::cout and nobody else's (the order-dependent test failure VERDICT r03 found).
void ref_mute_stdout(int mute) {
    if (mute) std::cout.setstate(std::ios_base::failbit);
    else std::cout.clear();
}

void* ref_setup(int nx, int nu, int N, const double* A, const double* B, const double* f,
                const double* Qdiag, const double* Rdiag, double rho) {
    RefHandle* h = new RefHandle();
    tinyMatrix Adyn = cm(A, nx, nx), Bdyn = cm(B, nx, nu);
    tinyVector fdyn = f ? tinyVector(cm(f, nx, 1)) : tinyVector(tinyVector::Zero(nx));
    tinyVector Q = cm(Qdiag, nx, 1), R = cm(Rdiag, nu, 1);
    // same call shape as examples/quadrotor_hovering.cpp:47-49
    int status = tiny_setup(&h->solver, Adyn, Bdyn, fdyn, Q.asDiagonal(), R.asDiagonal(), rho, nx, nu, N, 0);
    if (status) { delete h; return nullptr; }
    return h;
}

void ref_free(void* hv) { delete static_cast<RefHandle*>(hv); }  // the reference has no destroy API

int ref_set_bounds(void* hv, const double* x_min, const double* x_max, const double* u_min, const double* u_max) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    const int nx = s->work->nx, nu = s->work->nu, N = s->work->N;
    return tiny_set_bound_constraints(s, cm(x_min, nx, N), cm(x_max, nx, N), cm(u_min, nu, N - 1), cm(u_max, nu, N - 1));
}

// STATE triple first == the positional order of the definition (tiny_api.cpp:176-178).
int ref_set_cones(void* hv, int nsc, const int* Acx, const int* qcx, const double* cx,
                  int nic, const int* Acu, const int* qcu, const double* cu) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    VectorXi aAcx = Eigen::Map<const VectorXi>(Acx, nsc), aqcx = Eigen::Map<const VectorXi>(qcx, nsc);
    VectorXi aAcu = Eigen::Map<const VectorXi>(Acu, nic), aqcu = Eigen::Map<const VectorXi>(qcu, nic);
    tinyVector acx = cm(cx, nsc, 1), acu = cm(cu, nic, 1);
    return tiny_set_cone_constraints(s, aAcx, aqcx, acx, aAcu, aqcu, acu);
}

int ref_set_linear(void* hv, int nsl, const double* Ax, const double* bx, int nil, const double* Au, const double* bu) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    return tiny_set_linear_constraints(s, cm(Ax, nsl, s->work->nx), tinyVector(cm(bx, nsl, 1)),
                                       cm(Au, nil, s->work->nu), tinyVector(cm(bu, nil, 1)));
}

int ref_set_tv_linear(void* hv, int ntsl, const double* Ax, const double* bx, int ntil, const double* Au, const double* bu) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    const int nx = s->work->nx, nu = s->work->nu, N = s->work->N;
    return tiny_set_tv_linear_constraints(s, cm(Ax, ntsl * N, nx), cm(bx, ntsl, N), cm(Au, ntil * (N - 1), nu),
                                          cm(bu, ntil, N - 1));
}

double* ref_ptr(void* hv, const char* name, int* rows, int* cols) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    TinyWorkspace* w = s->work;
    TinyCache* c = s->cache;
    const std::string n(name);
    tinyMatrix* m = nullptr;
    tinyVector* v = nullptr;
#define M_(str, field) if (n == str) m = &(field);
#define V_(str, field) if (n == str) v = &(field);
    M_("Kinf", c->Kinf) M_("Pinf", c->Pinf) M_("Quu_inv", c->Quu_inv) M_("AmBKt", c->AmBKt)
    V_("APf", c->APf) V_("BPf", c->BPf)
    M_("Adyn", w->Adyn) M_("Bdyn", w->Bdyn) V_("fdyn", w->fdyn) V_("Q", w->Q) V_("R", w->R)
    M_("Xref", w->Xref) M_("Uref", w->Uref)
    M_("x", w->x) M_("u", w->u) M_("q", w->q) M_("r", w->r) M_("p", w->p) M_("d", w->d)
    M_("v", w->v) M_("vnew", w->vnew) M_("z", w->z) M_("znew", w->znew) M_("g", w->g) M_("y", w->y)
    M_("x_min", w->x_min) M_("x_max", w->x_max) M_("u_min", w->u_min) M_("u_max", w->u_max)
    M_("vc", w->vc) M_("vcnew", w->vcnew) M_("zc", w->zc) M_("zcnew", w->zcnew) M_("gc", w->gc) M_("yc", w->yc)
    M_("sol_x", s->solution->x) M_("sol_u", s->solution->u)
    M_("vlnew", w->vlnew) M_("zlnew", w->zlnew) M_("gl", w->gl) M_("yl", w->yl)
    M_("vlnew_tv", w->vlnew_tv) M_("zlnew_tv", w->zlnew_tv) M_("gl_tv", w->gl_tv) M_("yl_tv", w->yl_tv)
    M_("C1", c->C1) M_("C2", c->C2)
    M_("dKinf_drho", c->dKinf_drho) M_("dPinf_drho", c->dPinf_drho) M_("dC1_drho", c->dC1_drho) M_("dC2_drho", c->dC2_drho)
#undef M_
#undef V_
    if (m) { if (rows) *rows = (int)m->rows(); if (cols) *cols = (int)m->cols(); return m->data(); }
    if (v) { if (rows) *rows = (int)v->rows(); if (cols) *cols = 1; return v->data(); }
    return nullptr;
}

double ref_get(void* hv, const char* name) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    const std::string n(name);
    if (n == "abs_pri_tol") return s->settings->abs_pri_tol;
    if (n == "abs_dua_tol") return s->settings->abs_dua_tol;
    if (n == "max_iter") return s->settings->max_iter;
    if (n == "check_termination") return s->settings->check_termination;
    if (n == "en_state_bound") return s->settings->en_state_bound;
    if (n == "en_input_bound") return s->settings->en_input_bound;
    if (n == "en_state_soc") return s->settings->en_state_soc;
    if (n == "en_input_soc") return s->settings->en_input_soc;
    if (n == "en_state_linear") return s->settings->en_state_linear;
    if (n == "en_input_linear") return s->settings->en_input_linear;
    if (n == "en_tv_state_linear") return s->settings->en_tv_state_linear;
    if (n == "en_tv_input_linear") return s->settings->en_tv_input_linear;
    if (n == "primal_residual_state") return s->work->primal_residual_state;
    if (n == "primal_residual_input") return s->work->primal_residual_input;
    if (n == "dual_residual_state") return s->work->dual_residual_state;
    if (n == "dual_residual_input") return s->work->dual_residual_input;
    if (n == "status") return s->work->status;
    if (n == "iter") return s->work->iter;
    if (n == "sol_iter") return s->solution->iter;
    if (n == "sol_solved") return s->solution->solved;
    if (n == "rho") return s->cache->rho;
    if (n == "adaptive_rho") return s->settings->adaptive_rho;
    if (n == "adaptive_rho_min") return s->settings->adaptive_rho_min;
    if (n == "adaptive_rho_max") return s->settings->adaptive_rho_max;
    if (n == "adaptive_rho_enable_clipping") return s->settings->adaptive_rho_enable_clipping;
    if (n == "nx") return s->work->nx;
    if (n == "nu") return s->work->nu;
    if (n == "N") return s->work->N;
    return std::nan("");
}

int ref_set(void* hv, const char* name, double v) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    const std::string n(name);
    if (n == "abs_pri_tol") s->settings->abs_pri_tol = v;
    else if (n == "abs_dua_tol") s->settings->abs_dua_tol = v;
    else if (n == "max_iter") s->settings->max_iter = (int)v;
    else if (n == "check_termination") s->settings->check_termination = (int)v;
    else if (n == "en_state_bound") s->settings->en_state_bound = (int)v;
    else if (n == "en_input_bound") s->settings->en_input_bound = (int)v;
    else if (n == "en_state_soc") s->settings->en_state_soc = (int)v;
    else if (n == "en_input_soc") s->settings->en_input_soc = (int)v;
    else if (n == "en_state_linear") s->settings->en_state_linear = (int)v;
    else if (n == "en_input_linear") s->settings->en_input_linear = (int)v;
    else if (n == "en_tv_state_linear") s->settings->en_tv_state_linear = (int)v;
    else if (n == "en_tv_input_linear") s->settings->en_tv_input_linear = (int)v;
    else if (n == "adaptive_rho") s->settings->adaptive_rho = (int)v;
    else if (n == "adaptive_rho_min") s->settings->adaptive_rho_min = v;
    else if (n == "adaptive_rho_max") s->settings->adaptive_rho_max = v;
    else if (n == "adaptive_rho_enable_clipping") s->settings->adaptive_rho_enable_clipping = (int)v;
    else if (n == "rho") s->cache->rho = v;
    else return 1;
    return 0;
}

// Adaptive rho (admm.cpp:339-345, 397-423).  solve() declares `RhoAdapter adapter;` without initialising it, and
// format_matrices (rho_benchmark.cpp:57-59) sizes the adapter's matrices only when the indeterminate
// `adapter.matrices_initialized` happens to read false: with a nonzero byte there the 0x0 matrices are written out of
// bounds.  ref_stack_fill paints the stack region solve()'s frame is going to occupy, so that the flag's value is chosen
// by the caller instead of by whatever ran before: fill 0 = the deterministic behaviour the goldens pin ("matrices sized
// at the first adaptation of every solve"), any other byte = the probe of tools/adaptive_rho_probe.py.
__attribute__((noinline)) void ref_stack_fill(int byte, int bytes) {
    volatile char pad[1 << 16];
    const int n = bytes < (int)sizeof(pad) ? bytes : (int)sizeof(pad);
    for (int i = 0; i < n; ++i) pad[sizeof(pad) - 1 - i] = (char)byte;   // the top of this frame = where the next call's frame starts
    for (int i = 0; i < (int)sizeof(pad) - n; ++i) pad[i] = (char)byte;
}

// tiny_initialize_sensitivity_matrices (tiny_api.cpp:479-540): quadrotor-sized tables only (4x12 / 12x12 / 4x4)
int ref_init_sensitivity(void* hv) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    if (s->work->nx != 12 || s->work->nu != 4) return 1;
    tiny_initialize_sensitivity_matrices(s);
    return 0;
}

// other shapes than the quadrotor's: size the four tables (zero), the caller fills them through ref_ptr
void ref_alloc_sensitivity(void* hv) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    const int nx = s->work->nx, nu = s->work->nu;
    s->cache->dKinf_drho = tinyMatrix::Zero(nu, nx);
    s->cache->dPinf_drho = tinyMatrix::Zero(nx, nx);
    s->cache->dC1_drho = tinyMatrix::Zero(nu, nu);
    s->cache->dC2_drho = tinyMatrix::Zero(nx, nx);
}

int ref_solve(void* hv) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    if (s->settings->adaptive_rho) ref_stack_fill(0, 1 << 16);
    return tiny_solve(s);
}
// the probe's variants: paint the stack with `byte` right before the solve (nothing runs in between), or not at all (< 0)
int ref_solve_fill(void* hv, int byte) {
    if (byte >= 0) ref_stack_fill(byte, 1 << 16);
    return tiny_solve(static_cast<RefHandle*>(hv)->solver);
}

int ref_phase(void* hv, const char* name) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    const std::string n(name);
    if (n == "update_linear_cost") { update_linear_cost(s); return 0; }
    if (n == "backward_pass_grad") { backward_pass_grad(s); return 0; }
    if (n == "forward_pass") { forward_pass(s); return 0; }
    if (n == "update_slack") { update_slack(s); return 0; }
    if (n == "update_dual") { update_dual(s); return 0; }
    if (n == "termination_condition") return termination_condition(s) ? 1 : 0;
    return -1;
}

void ref_project_soc(double* sv, int n, double mu) {
    tinyVector s = cm(sv, n, 1);
    tinyVector r = project_soc(s, (float)mu);   // the reference's parameter is a float (admm.hpp:24)
    for (int i = 0; i < n; ++i) sv[i] = r(i);
}

long ref_closed_loop(void* hv, double* x0, int steps, int* iters_out, double* u0_out) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    TinyWorkspace* w = s->work;
    const int nx = w->nx, nu = w->nu;
    tinyVector x = cm(x0, nx, 1);
    long total = 0;
    for (int k = 0; k < steps; ++k) {
        tiny_set_x0(s, x);
        if (s->settings->adaptive_rho) ref_stack_fill(0, 1 << 16);
        tiny_solve(s);
        total += s->solution->iter;
        if (iters_out) iters_out[k] = s->solution->iter;
        if (u0_out) std::memcpy(u0_out + (size_t)nu * k, w->u.data(), sizeof(double) * nu);
        x = w->Adyn * x + w->Bdyn * w->u.col(0) + w->fdyn;
    }
    std::memcpy(x0, x.data(), sizeof(double) * nx);
    return total;
}

// the same with a moving state-reference window (see oracle_closed_loop_traj): examples/rocket_landing_mpc.cpp:120-135
long ref_closed_loop_traj(void* hv, double* x0, int steps, const double* traj, int points, int k0, int* iters_out, double* u0_out) {
    TinySolver* s = static_cast<RefHandle*>(hv)->solver;
    TinyWorkspace* w = s->work;
    const int nx = w->nx, nu = w->nu, N = w->N;
    tinyVector x = cm(x0, nx, 1);
    long total = 0;
    for (int k = 0; k < steps; ++k) {
        w->x.col(0) = x;
        for (int i = 0; i < N; ++i) {
            int kk = k0 + k + i;
            if (kk > points - 1) kk = points - 1;
            w->Xref.col(i) = cm(traj + (size_t)nx * kk, nx, 1);
        }
        tiny_solve(s);
        total += s->solution->iter;
        if (iters_out) iters_out[k] = s->solution->solved ? s->solution->iter : -s->solution->iter;
        if (u0_out) std::memcpy(u0_out + (size_t)nu * k, w->u.data(), sizeof(double) * nu);
        x = w->Adyn * x + w->Bdyn * w->u.col(0) + w->fdyn;
    }
    std::memcpy(x0, x.data(), sizeof(double) * nx);
    return total;
}

}  // extern "C"
