/*
 * tinympc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (no Eigen) CPU restatement of the TinyMPC ADMM hot path, used ONLY as
 * the parity checker for the MI355X HIP implementation and as the "port" CPU
 * baseline of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call anything in this directory.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * TinyMPC checkout, /root/reference):
 *   src/tinympc/admm.cpp      (solve loop + phases + projections)
 *   src/tinympc/tiny_api.cpp  (setup, cache precompute, setters)
 *
 * Parity pinning: the reference ships no tests or golden vectors (SURVEY.md
 * section 4), so this restatement is pinned against outputs of the reference
 * itself: oracle/_ref/libtinympc_ref.so is the real reference compiled from
 * /root/reference by oracle/Makefile, oracle/gen_golden.py drives it and commits
 * its outputs under tests/golden/, and tests/test_oracle_*.py check this file
 * against those fixtures (and against the live _ref library when present).
 *
 * All matrices are column-major (Eigen default), sizes as in
 * src/tinympc/types.hpp:88-208.
 */
#ifndef TINYMPC_ORACLE_H
#define TINYMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleSolver OracleSolver;

/* tiny_setup (tiny_api.cpp:21-147): Qdiag/Rdiag are the USER diagonals (without
 * rho); the double-rho of the cache path (tiny_api.cpp:117-118,136,317-318) is
 * reproduced inside.  Returns NULL on bad dimensions. */
OracleSolver* oracle_setup(int nx, int nu, int N,
                           const double* A, const double* B, const double* f,
                           const double* Qdiag, const double* Rdiag, double rho);
void oracle_free(OracleSolver* s);

/* tiny_set_bound_constraints (tiny_api.cpp:149-174): nx*N, nx*N, nu*(N-1), nu*(N-1) */
int oracle_set_bounds(OracleSolver* s, const double* x_min, const double* x_max,
                      const double* u_min, const double* u_max);
/* tiny_set_cone_constraints (tiny_api.cpp:176-208), STATE triple first (the
 * definition's positional order). */
int oracle_set_cones(OracleSolver* s,
                     int n_state_cones, const int* Acx, const int* qcx, const double* cx,
                     int n_input_cones, const int* Acu, const int* qcu, const double* cu);

/* tiny_set_linear_constraints (tiny_api.cpp:210-251) / tiny_set_tv_linear_constraints (:253-304):
 * half-spaces a_k' z <= b_k; A matrices column-major, one ROW per constraint; the time-varying
 * counts are per knot point (tv_Alin_x is (n*N) x nx, tv_blin_x n x N, ...). */
int oracle_set_linear(OracleSolver* s, int n_state, const double* Alin_x, const double* blin_x,
                      int n_input, const double* Alin_u, const double* blin_u);
int oracle_set_tv_linear(OracleSolver* s, int n_state, const double* tv_Alin_x, const double* tv_blin_x,
                         int n_input, const double* tv_Alin_u, const double* tv_blin_u);

/* Named access to every matrix/vector of TinyCache / TinyWorkspace / TinySolution
 * (names = reference field names; "sol_x"/"sol_u" for TinySolution).  Returns
 * NULL for unknown names. rows/cols may be NULL. */
double* oracle_ptr(OracleSolver* s, const char* name, int* rows, int* cols);
/* Named scalar access: settings (abs_pri_tol, abs_dua_tol, max_iter,
 * check_termination, en_state_bound, en_input_bound, en_state_soc, en_input_soc,
 * en_state_linear, en_input_linear, en_tv_state_linear, en_tv_input_linear),
 * status (primal_residual_state, ..., status, iter, sol_iter, sol_solved),
 * rho, riccati_iters. */
double oracle_get(OracleSolver* s, const char* name);
int oracle_set(OracleSolver* s, const char* name, double value);

/* admm.cpp:331-455 */
int oracle_solve(OracleSolver* s);
/* Individual phases (admm.cpp:13-32, 81-135, 219-235, 262-297, 310-328);
 * name in {update_linear_cost, backward_pass_grad, forward_pass, update_slack,
 * update_dual, termination_condition}; returns the bool for termination_condition,
 * 0 otherwise, -1 for unknown names. */
int oracle_phase(OracleSolver* s, const char* name);
/* project_soc (admm.cpp:39-60) on an n-vector in place (n must be 3, as in the reference). */
void oracle_project_soc(double* s, int n, double mu);

/* The MPC closed loop of the examples (examples/quadrotor_hovering.cpp:73-93):
 * steps x { x[:,0] = x0; solve; x0 = A x0 + B u[:,0] + f }.  iters_out[steps],
 * u0_out[steps*nu] may be NULL. x0 is updated in place. Returns total iterations. */
long oracle_closed_loop(OracleSolver* s, double* x0, int steps, int* iters_out, double* u0_out);
long oracle_closed_loop_traj(OracleSolver* s, double* x0, int steps, const double* traj, int points, int k0, int* iters_out, double* u0_out);

#ifdef __cplusplus
}
#endif
#endif
