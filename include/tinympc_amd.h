/*
 * tinympc_amd.h -- C ABI of the MI355X-native batched TinyMPC ADMM solver (libtinympc_amd.so).
 *
 * Two surfaces, both plain C (pointers and sizes only, no C++/Eigen/torch types):
 *
 *  (A) the BATCHED, device-resident surface (tiny_batch_*): what the throughput metric is measured
 *      on.  It mirrors the reference's operator interface for the hot path one-for-one --
 *      tiny_setup / tiny_set_bound_constraints / tiny_set_cone_constraints / tiny_update_settings /
 *      tiny_set_x0 / tiny_set_x_ref / tiny_set_u_ref / tiny_solve  (src/tinympc/tiny_api.hpp:10-54)
 *      -- but every per-instance quantity carries a leading batch axis and lives in HBM.
 *
 *  (B) the reference's own entry points (tiny_setup, tiny_solve, ... same names, same argument
 *      meaning, same return codes) over plain-data mirrors of the reference structs, so existing
 *      callers compiled against the reference's headers link against this library unchanged
 *      (INTEGRATION.md).  tiny_solve(TinySolver*) is a batch of one on the GPU.
 *
 * There is NO CPU fallback: every solve runs the HIP kernels; without a GPU the calls fail with
 * TINY_ERR_NO_DEVICE.
 *
 * Matrix conventions: column-major (Eigen default).  Per-instance matrices of a batch are stored
 * back to back: Xref is [batch][N][nx] doubles (instance-major, then knot point, then row).
 */
#ifndef TINYMPC_AMD_H
#define TINYMPC_AMD_H

#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* return codes: 0 / 1 keep the reference's meaning (tiny_solve: 0 = all converged, 1 = some
 * instance hit max_iter, admm.cpp:441,454; setters: 0 = ok, 1 = dimension error).              */
#define TINY_OK                 0
#define TINY_ERR_DIM            1
#define TINY_ERR_NULL          (-1)
#define TINY_ERR_NO_DEVICE     (-2)   /* no MI355X / HIP runtime failure: there is no CPU path   */
#define TINY_ERR_UNSUPPORTED   (-3)   /* (nx,nu,N) not instantiated, or an out-of-scope feature   */
#define TINY_ERR_HIP           (-4)
#define TINY_ERR_ARG           (-5)

/* work->status values (admm.cpp:336,431) */
#define TINY_STATUS_SOLVED      1
#define TINY_STATUS_UNSOLVED   11

typedef struct TinyBatch TinyBatch;   /* opaque, owns its HBM buffers */

/* Per-instance fields (reference names, src/tinympc/types.hpp:88-208). State-sized = nx x N,
 * input-sized = nu x (N-1). */
typedef enum {
    TINY_F_X0 = 0,      /* nx            tiny_set_x0  (x[:,0])                */
    TINY_F_XREF,        /* nx x N        tiny_set_x_ref                        */
    TINY_F_UREF,        /* nu x (N-1)    tiny_set_u_ref                        */
    TINY_F_X,           /* work->x      (primal; col 0 is x0)                 */
    TINY_F_U,           /* work->u                                             */
    TINY_F_VNEW,        /* work->vnew   (== solution->x after a solve)         */
    TINY_F_ZNEW,        /* work->znew   (== solution->u)                       */
    TINY_F_G,           /* work->g                                             */
    TINY_F_Y,           /* work->y                                             */
    TINY_F_V,           /* work->v                                             */
    TINY_F_Z,           /* work->z                                             */
    TINY_F_VCNEW,       /* work->vcnew  (cone slack, only with a cone enabled) */
    TINY_F_ZCNEW,
    TINY_F_GC,
    TINY_F_YC,
    TINY_F_Q,           /* work->q / r / p / d of the last iteration: kept by  */
    TINY_F_R,           /* a solve after tiny_batch_set_option("debug", 1);    */
    TINY_F_P,           /* settable as inputs of tiny_batch_phase              */
    TINY_F_D,
    TINY_F_VLNEW,       /* work->vlnew / zlnew / gl / yl: static linear-constraint slack + dual   */
    TINY_F_ZLNEW,
    TINY_F_GL,
    TINY_F_YL,
    TINY_F_VLNEW_TV,    /* work->vlnew_tv / zlnew_tv / gl_tv / yl_tv: time-varying linear          */
    TINY_F_ZLNEW_TV,
    TINY_F_GL_TV,
    TINY_F_YL_TV,
    TINY_F_COUNT
} TinyField;

/* flags for tiny_batch_set / tiny_batch_get */
#define TINY_HOST        0   /* pointer is host memory, [batch] matrices back to back          */
#define TINY_DEVICE      1   /* pointer is device (HBM) memory on the batch's GPU               */
#define TINY_BROADCAST   2   /* (set only) ONE matrix, replicated to every instance              */

/* ---- problem family ---------------------------------------------------------------------- */
int tiny_batch_device_count(void);

/* == tiny_setup (tiny_api.hpp:10-12, tiny_api.cpp:21-147) for `batch` instances sharing one
 * (A, B, f, Q, R, rho).  Qdiag/Rdiag are the USER diagonals (the reference's callers pass
 * Q.asDiagonal()); rho is added the way the reference does (twice on the cache path).
 * Runs the infinite-horizon Riccati recursion on the host (tiny_api.cpp:307-381), allocates the
 * HBM records (zero-initialised, like tiny_setup) on GPU `device`.                                */
int tiny_batch_setup(TinyBatch** out, const double* Adyn, const double* Bdyn, const double* fdyn,
                     const double* Qdiag, const double* Rdiag, double rho,
                     int nx, int nu, int N, int batch, int device, int verbose);
/* Heterogeneous batch: instance i has its OWN (A_i, B_i, f_i, Q_i, R_i, rho_i) -- arrays with a leading batch
 * axis ([batch][nx*nx] column-major, [batch][nx*nu], [batch][nx] (may be NULL), [batch][nx], [batch][nu], [batch]).
 * tiny_precompute_and_set_cache (tiny_api.cpp:307-381) runs on the GPU for all instances at once; the solve kernel
 * then streams each instance's own cache.  Bounds, cones and settings stay shared.  Any nx + nu <= 32, nu <= 16 (round 5): shapes the
 * one-row kernel holds run its HET variant, wide and long shapes the tile kernel's per-instance form on the shape's fastest box form
 * (run-time instantiated: needs hipRTC or the prebuilt store); TINY_ERR_UNSUPPORTED when neither kernel holds the shape.  Round 6: cones
 * that share rows, more half-spaces per knot than the register variants hold, or a variant hipRTC cannot make send a heterogeneous
 * batch to the coverage kernel, which reads the same per-instance tables (adaptive rho does not combine with per-instance data). */
int tiny_batch_setup_hetero(TinyBatch** out, const double* Adyn, const double* Bdyn, const double* fdyn,
                            const double* Qdiag, const double* Rdiag, const double* rho,
                            int nx, int nu, int N, int batch, int device, int verbose);
/* one instance's cache member (names as tiny_batch_get_cache, plus "riccati_iters") */
int tiny_batch_get_cache_instance(TinyBatch* b, int instance, const char* name, double* out, int capacity);
int tiny_batch_destroy(TinyBatch* b);

/* == tiny_set_bound_constraints (tiny_api.hpp:13-15): nx x N, nx x N, nu x (N-1), nu x (N-1),
 * column-major, shared by every instance. */
int tiny_batch_set_bound_constraints(TinyBatch* b, const double* x_min, const double* x_max,
                                     const double* u_min, const double* u_max);
/* == tiny_set_cone_constraints (tiny_api.cpp:176-208): STATE triple first (the definition's
 * positional order).  Cones have dimension 3 (admm.cpp:53).  Cones of one family may share rows: the reference projects them
 * one after the other (admm.cpp:111-135) and so does this library -- on the coverage kernel (single-step launches, shared
 * problem data); disjoint cones run on the register-resident kernels. */
int tiny_batch_set_cone_constraints(TinyBatch* b, int n_state_cones, const int* Acx, const int* qcx,
                                    const double* cx, int n_input_cones, const int* Acu,
                                    const int* qcu, const double* cu);
/* == tiny_set_linear_constraints (tiny_api.hpp:19-21, tiny_api.cpp:210-251): half-spaces a_k' z <= b_k,
 * Alin_x is n_state x nx column-major (one ROW per constraint), blin_x n_state; same for the inputs. */
int tiny_batch_set_linear_constraints(TinyBatch* b, int n_state, const double* Alin_x, const double* blin_x,
                                      int n_input, const double* Alin_u, const double* blin_u);
/* == tiny_set_tv_linear_constraints (tiny_api.hpp:22-24, tiny_api.cpp:253-304): counts are PER KNOT POINT;
 * tv_Alin_x is (n_state*N) x nx column-major (row n_state*i + k = constraint k at knot i), tv_blin_x
 * n_state x N; tv_Alin_u (n_input*(N-1)) x nu, tv_blin_u n_input x (N-1). */
int tiny_batch_set_tv_linear_constraints(TinyBatch* b, int n_state, const double* tv_Alin_x, const double* tv_blin_x,
                                         int n_input, const double* tv_Alin_u, const double* tv_blin_u);
/* == tiny_update_settings (tiny_api.hpp:36-42).  With a linear / time-varying-linear switch on (or a shape
 * without a register-resident instantiation) the solve runs the coverage kernel (general_kernel.hip.h). */
int tiny_batch_update_settings(TinyBatch* b, double abs_pri_tol, double abs_dua_tol, int max_iter,
                               int check_termination, int en_state_bound, int en_input_bound,
                               int en_state_soc, int en_input_soc, int en_state_linear,
                               int en_input_linear, int en_tv_state_linear, int en_tv_input_linear);
/* == settings->adaptive_rho, adaptive_rho_min, adaptive_rho_max, adaptive_rho_enable_clipping (types.hpp:75-79).  With it on,
 * every 5th ADMM iteration re-estimates rho from the residuals of an OSQP-style restatement of the problem and moves Kinf,
 * Pinf (and the dead copies C1, C2) by a first-order Taylor step, as admm.cpp:397-423 / rho_benchmark.cpp:14-249 do -- NOT
 * Quu_inv, AmBKt, APf, BPf, Q, R, which the reference keeps as they were.  The cache then is per-instance STATE that persists
 * from solve to solve (tiny_batch_reset puts tiny_setup's cache back).  The reference's RhoAdapter flag is read uninitialised
 * upstream; the behaviour reproduced is the one with that flag false ("matrices sized by the first adaptation of every
 * solve"), which is also the only one that does not crash (profiles/r02_adaptive_rho_probe.txt).  One-row kernel only. */
int tiny_batch_set_adaptive_rho(TinyBatch* b, int enable, double rho_min, double rho_max, int enable_clipping);
/* == cache->dKinf_drho (nu x nx), dPinf_drho (nx x nx), dC1_drho (nu x nu), dC2_drho (nx x nx): column-major, shared by every
 * instance; dC1 / dC2 may be NULL (zero).  tiny_initialize_sensitivity_matrices (tiny_api.cpp:479-540) is the quadrotor's set. */
int tiny_batch_set_sensitivity(TinyBatch* b, const double* dKinf_drho, const double* dPinf_drho, const double* dC1_drho,
                               const double* dC2_drho);
/* per-instance cache state of an adaptive batch, host arrays with a leading batch axis, column-major matrices: which =
 * "rho" [batch], "Kinf" [batch][nu*nx], "Pinf" [batch][nx*nx], "C1" [batch][nu*nu], "C2" [batch][nx*nx] */
int tiny_batch_set_cache_state(TinyBatch* b, const char* which, const double* src);
int tiny_batch_get_cache_state(TinyBatch* b, const char* which, double* dst);
/* Read back one cache matrix (TinyCache, types.hpp:43-59) by name: "Kinf" (nu x nx), "Pinf",
 * "Quu_inv", "AmBKt", "APf", "BPf", or "Q"/"R" (work->Q/R = user + rho).  Returns element count. */
int tiny_batch_get_cache(TinyBatch* b, const char* name, double* out, int capacity);
/* Overwrite one of them (same names and sizes, plus "rho": one double): a caller's own cache instead of the Riccati recursion of
 * tiny_batch_setup -- what the code generated by tiny_codegen uses to restore a frozen TinyCache. */
int tiny_batch_set_cache(TinyBatch* b, const char* name, const double* src);

/* ---- per-instance data ------------------------------------------------------------------- */
/* tiny_set_x0 / tiny_set_x_ref / tiny_set_u_ref and direct workspace pokes of the examples
 * (work->Xref = ..., work->y = 0, examples/quadrotor_tracking.cpp:89-93). */
int tiny_batch_set(TinyBatch* b, TinyField field, const double* src, int flags);
int tiny_batch_get(TinyBatch* b, TinyField field, double* dst, int flags);
/* zero every warm-start record (x,u,vnew,znew,v,z,g,y,vcnew,zcnew,gc,yc) = state after tiny_setup */
int tiny_batch_reset(TinyBatch* b);

/* ---- the hot path ------------------------------------------------------------------------ */
/* == tiny_solve (tiny_api.hpp:34): one ADMM solve of every instance from its warm state.
 * Synchronous; returns 0 when every instance converged, 1 otherwise (admm.cpp:441,454). */
int tiny_batch_solve(TinyBatch* b);
/* Same launch, asynchronous on the batch's stream; returns TINY_OK after enqueueing. */
int tiny_batch_solve_async(TinyBatch* b);
int tiny_batch_synchronize(TinyBatch* b);
/* ONE phase of the ADMM iteration over the whole batch, on the device records as they are (asynchronous): the
 * batched form of the phase functions the reference exports (src/tinympc/admm.hpp:12-17).  Reads / writes exactly
 * the workspace fields the reference function does (x|u, q|r, p|d, the slack and dual families). */
enum {
    TINY_PHASE_LINEAR_COST = 1,   /* update_linear_cost     admm.cpp:262-304 */
    TINY_PHASE_BACKWARD = 2,      /* backward_pass_grad     admm.cpp:13-20   */
    TINY_PHASE_FORWARD = 3,       /* forward_pass           admm.cpp:25-32   */
    TINY_PHASE_SLACK = 4,         /* update_slack           admm.cpp:81-211  */
    TINY_PHASE_DUAL = 5,          /* update_dual            admm.cpp:219-256 */
    TINY_PHASE_TERMINATION = 6    /* termination_condition  admm.cpp:310-328: the four residuals (always evaluated);
                                     tiny_batch_get_status' `solved` array then holds the returned bool */
};
int tiny_batch_phase(TinyBatch* b, int phase);
/* solution->iter, solution->solved, work->status, and the four residuals
 * {primal_state, primal_input, dual_state, dual_input} ([batch][4]); any pointer may be NULL. */
int tiny_batch_get_status(TinyBatch* b, int* iter, int* solved, int* status, double* residuals);
/* Device-side reduction over the batch: out[0..9] = {sum iter, sum solved, batch,
 * max primal_state, max primal_input, max dual_state, max dual_input,
 * iterations accumulated since the last tiny_batch_reset, converged solves accumulated, 0}.
 * host_out and/or device_out (a device pointer to 10 doubles, e.g. the buffer handed to an
 * RCCL collective) may be NULL. */
int tiny_batch_reduce_stats(TinyBatch* b, double* host_out, void* device_out);

/* ---- plumbing ---------------------------------------------------------------------------- */
/* Options: "advance_x0" (1: each solve also writes x0 <- A x0 + B u[:,0] + f, the plant step of
 * the examples' closed loop, examples/quadrotor_hovering.cpp:92), "debug" (1: keep q,r,p,d),
 * "grid_waves_per_cu" (persistent-grid size, 0 = one wave per tile), "dpp_mode" (2 [default] fused
 * v_fmac_f64_dpp on one accumulator chain, 0 the same on two chains, 1 v_mov_dpp + v_fma), "timing" (n: record HIP events for the next n solves),
 * "prefer_tile" / "no_tile" (experiments: take / avoid the tile kernel where a choice exists),
 * "force_general" (1: use the coverage kernel even when a
 * register-resident instantiation exists), "steps_per_launch" (T >= 1: every solve call runs T closed-loop MPC steps -- solve, plant step
 * x0 <- A x0 + B u[:,0] + f, solve, ... -- inside ONE launch with the ADMM state held in registers;
 * references stay fixed during the launch), "step_log" (1: keep per-step iteration counts / u0),
 * "reset_duals" (1: g = 0, y = 0 before every solve, examples/quadrotor_tracking.cpp:92-93), "traj_step",
 * "one_shot" (one-shot / cold solves: the warm-start state is taken as zero -- the state after tiny_setup or
 * tiny_batch_reset -- WITHOUT being read, and only the results are written: 1 = x|u and vnew|znew (solution->x|u),
 * 2 = x|u only, the bytes_cold = 8(nx+2S)+44 traffic of a solve that is not going to be warm-started; the other
 * warm-start records are left as they were.  Round 5: every shape -- wide and long ones on the tile kernel's EXT forms, as are
 * reference windows, "reset_duals" and per-instance problem data; the coverage kernel (overlapping cones, no hipRTC) runs these launch
 * forms and fused steps too, as a loop of single-step launches, and then writes every record back),
 * "repack_tail" (default -1 = 0; 1: the open instances of a split solve's capped first stage run to max_iter in ONE launch of the tile
 * kernel's dynamic slot form instead of the follow-up stages; bit-identical; measured slower than the staged lists under the default
 * dispatch -- profiles/r06_negative_results.md -- and therefore off; read-back "last_tail_tile"),
 * "one_shot_fast" (default 1, round 6: a one-shot launch of a wide / long shape rides on the shape's fast box form -- LDS-offload set,
 * dynamic slots; a form that streams v|z streams into a scratch array of the record's shape, allocated on first use, so that the
 * record stays untouched -- 1.2-1.9x the rate of the all-in-registers form, which 0 brings back; profiles/r06_one_shot_tile_forms.md),
 * "plan" (default 1: a fresh handle whose shape, settings and batch bucket match an entry of tinympc_amd/data/plans.txt -- or of the
 * file TINYMPC_AMD_PLANS names; "0": none -- takes that entry's launch form on its FIRST solve instead of probing; read-back
 * "plan_shipped"; tools/make_plans.py writes the file),
 * "het_ub" (default 1: per-instance problem data with a knot-invariant box takes the variant that keeps the box in two registers),
 * "repack_after" (-1, the default: automatic -- K is derived from the iteration histogram of the batch's previous solve by a
 * cost model, 0 (a plain launch) when the counts are uniform enough that splitting would not pay 5 %; 0: never split;
 * K > 0: split solve for batches whose iteration counts diverge -- the launch stops at iteration K, the
 * instances still open are listed by the kernel itself and carried on to 2K, 4K, ... max_iter by follow-up launches, four
 * open instances per wave at every stage, so that a slow instance no longer holds a wave by itself.  Results are
 * bit-identical to the unsplit solve; K is rounded down to a multiple of check_termination.  Worth it for cold solves
 * with a tail of slow instances (config 3: -16 %), not for warm MPC steps (every stage is a launch).  One-row kernel,
 * single-step launches without "advance_x0" / "one_shot"; ignored elsewhere.  "repack_growth" (default 2) and
 * "repack_waves_per_cu" (default 8) tune the stage schedule and the grid of the follow-up launches),
 * "store_primal" (default 1; 0: launches do not write work->x|u back -- between closed-loop steps with "advance_x0" it has
 * no consumer, the plant step and solution->x|u = vnew|znew are still written; tiny_batch_get(TINY_F_X / TINY_F_U) then
 * returns stale data; 2: only the first knot is written -- x[:,0], x[:,1] and u[:,0], the control a closed-loop caller applies.
 * Ignored while a cone / half-space family is enabled, whose slack the next solve initialises from x|u).
 * "share_ref" (default 1: when Xref and Uref were set with TINY_BROADCAST -- or never -- every instance reads ONE reference
 * record instead of its own copy; 0 switches that off).
 * "launch_order" (default 1: successive plain launches of the register kernel walk the batch in ALTERNATING directions, so that a
 * warm launch begins with the records its predecessor touched last -- the ones the 256 MiB Infinity Cache in front of HBM still
 * holds; 0: always ascending; 2: always descending.  Instances are independent: no result depends on it.  With the same aim the box
 * variants store work->x|u -- written once per launch, never read back by a kernel -- with nontemporal stores).
 * "half_rows" (default -1 / 1: shapes with nx + nu <= 8 run TWO instances per 16-lane row, eight per wavefront, where that form is
 * compiled in -- bit-identical to the one-instance-per-row form; 0: off; read-back "last_half_rows").
 * "prefetch" (round 6; default -1: a plain single-step launch of the register kernel's box variant over a batch that gives every
 * resident wavefront at least three tiles -- and the first stage of a split solve -- runs as PERSISTENT waves that draw their tiles
 * from a ticket counter and receive the NEXT tile's warm-start / reference records by LDS-DMA while the current tile iterates: at
 * two waves per SIMD a wave's load phase otherwise hides behind one neighbour only; 0: never; 1: wherever the form exists, any
 * batch size.  Same instructions per iteration, instances independent: bit-identical.  "prefetch_static" (default -1 = by rule, 75 for warm launches and 50 for cold ones: percent of
 * a wave's tiles it takes by grid stride before it draws tickets), "prefetch_waves" (default 0 = what is resident; > 0: cap on the
 * persistent grid); read-back "last_prefetch").
 * "step_regroup" (fused launches, "steps_per_launch" > 1, of the register kernel.  The four rows of a wavefront run in lock step: a
 * wave costs what its slowest row costs.  -1, the default: when the iteration totals of the batch's previous fused launch say that
 * this costs >= 5 %, the launch runs as STRETCHES of K = steps / 4 (at least 8) MPC steps, each over the instances ordered by the
 * iteration count of their last solve -- a counting sort on the device between the stretches; K > 0: stretches of K steps; 0: never.
 * Every stretch is the launch steps_per_launch = K would have made: bit-identical results, step logs included.  Rocket landing with
 * the thrust cone, 65 536 instances x 90 steps: 30.2 -> 27.0 ms; a batch of identical instances is never cut.
 * "step_regroup_streams" (default 2: the two halves of the batch on two streams, half a stretch out of step, joined back into the
 * batch's stream inside the solve call; 1: one stream).  Read-backs "step_regroup_stretches", "step_regroup_verdict",
 * "lockstep_permille"),
 * "repack_sort" (split solves: -1, the default: a follow-up stage predicted to run >= 60 us takes its list of open instances ordered
 * by residual / tolerance, largest first -- instances that are equally far from converging share a wave; 1: every stage; 0: never.
 * Bit-identical; (12,2,30) x 131 072: 12.6 -> 10.9 ms.  Read-back "repack_sorted_stages").
 * A solve that converges at its first termination check never stores v|z: the reference returns before v = vnew. */
int tiny_batch_set_option(TinyBatch* b, const char* name, long value);
/* derived state: "auto_split_k" (the K the automatic split solve derived from the last iteration histogram; 0 = plain launch),
 * "auto_split_permille" (its predicted time, in 1/1000 of the plain launch's), "repack_after" */
long tiny_batch_get_option(TinyBatch* b, const char* name);
/* The settled launch form of a batch as plain data.  The library decides by the clock how a batch is launched (plain or split solve
 * and its K / stage schedule, the tile kernel's dynamic form for a one-row shape, stretches of MPC steps for a fused launch); the first
 * solves of a batch are probes.  tiny_batch_get_plan exports what they found, tiny_batch_set_plan imports it into another handle of
 * the same (nx, nu, N) -- in this process or another one: the plan is a POD without pointers, write it to a file as it is -- which
 * then takes the settled form on its FIRST solve.  A plan is advice about launch forms only: results are bit-identical whatever it
 * says.  It applies to solves with the max_iter it was learnt for (auto_cap_max_iter).  open_questions > 0: the exporter had not
 * finished probing (the importer carries on where it stopped). */
#define TINY_PLAN_MAGIC   0x4e4c5054   /* "TPLN" */
#define TINY_PLAN_VERSION 1
typedef struct TinyBatchPlan {
    int magic, version, bytes;
    int nx, nu, N, batch, max_iter, check_termination;
    int open_questions;
    int auto_verdict, auto_cap, auto_cap_max_iter, auto_growth, growth_verdict, auto_probes;   /* split solve: 1 kept / -1 rejected, K, its max_iter, stage growth 2 | 4 */
    int tile_verdict, regroup_verdict, hist_valid;                                             /* tile alternative / stretches of MPC steps: 1 on, -1 off, 0 open */
    double auto_plain_rate, auto_split_rate, auto_gain, tile_rate, lockstep_ratio;             /* the clock readings behind the verdicts (ms per instance-iteration) */
    unsigned hist[1024];                                                                       /* iteration histogram the schedule came from */
} TinyBatchPlan;
int tiny_batch_get_plan(TinyBatch* b, TinyBatchPlan* out);
int tiny_batch_set_plan(TinyBatch* b, const TinyBatchPlan* in);
/* the cost model behind "repack_after" = -1 by itself (host arithmetic, no GPU): hist[1024], hist[i] = instances whose solve takes
 * i iterations; returns the proposed K (0 = plain launch), *ratio = predicted time of the best split / of the plain launch */
int tiny_predict_split(const unsigned* hist, int nx, int nu, int N, int max_iter, int check_termination, int num_cus, double* ratio);
/* the schedule behind "step_regroup" by itself (host arithmetic, no GPU): the stretches a fused launch of `steps` MPC steps is cut into
 * -- k > 0: stretches of k steps, k <= 0: the automatic length (steps / 4, at least 8); known = 0: nothing is known about the instances
 * yet, ONE step of all of them goes first; half = 1: the second half of the batch under "step_regroup_streams" = 2, half a stretch out
 * of step with the first.  Writes the lengths to out[0 .. capacity), returns their number (their sum is `steps`). */
int tiny_step_regroup_plan(int steps, int k, int known, int half, int* out, int capacity);
int tiny_batch_set_stream(TinyBatch* b, void* hip_stream);      /* run on a caller-owned stream */
/* Closed-loop tracking (examples/quadrotor_tracking.cpp:65,89): a reference trajectory of n_points state
 * vectors ([n_points][nx] doubles), shared by every instance.  While it is set, the state reference of a solve
 * is the N-knot window starting at (MPC step counter + offsets[instance]); the counter starts at 0, advances by
 * one per MPC step (also inside fused launches) and can be moved with set_option("traj_step", k).  offsets may be
 * NULL; xref_points == NULL switches back to the per-instance Xref records.
 * What is left in the records DIFFERS by kernel path (tiny_batch_kernel_path tells which one runs):
 *   - register-resident kernels (one-row, tile): the window lives in registers; the Xref records keep what the caller set, and
 *     "one_shot" writes only the records its store mask names (g|y, v|z stay as they were);
 *   - coverage kernel (overlapping cones, no hipRTC): every windowed launch WRITES the window of its last MPC step into the Xref
 *     records (they are per-instance from then on: after set_reference_trajectory(NULL) they hold that window, not the caller's
 *     earlier reference -- set Xref again), and a "one_shot" launch rewrites EVERY warm-start record (prim, slack, dual,
 *     slack_prev, cone and half-space slacks) with the state the solve ended in.
 * A caller that leaves one_shot / windowed mode and must not depend on the path resets the batch (tiny_batch_reset) or sets Xref
 * and the warm-start fields it relies on. */
int tiny_batch_set_reference_trajectory(TinyBatch* b, const double* xref_points, int n_points, const int* offsets, int flags);
/* After a launch with "steps_per_launch" = T > 1 and "step_log" = 1: per fused MPC step and instance,
 * iters[T][batch] (negative = that solve hit max_iter) and the applied control u0[T][batch][nu]. */
int tiny_batch_get_step_log(TinyBatch* b, int* iters, double* u0, int steps);
/* kernel durations (ms) of the solves recorded since "timing" was set; returns the count */
int tiny_batch_get_timing(TinyBatch* b, float* ms, int capacity);
const char* tiny_batch_last_error(TinyBatch* b);
/* (nx,nu,N) kernel instantiations compiled into the library: writes up to capacity triples, returns the count */
int tiny_batch_supported_dims(int* triples, int capacity);
/* which kernel the next solve runs: 0 one-row register-resident kernel (admm_kernel.hip.h), 1 tile kernel
 * (tile_kernel.hip.h: wide / long shapes, W x R DPP rows per instance), 2 coverage kernel (general_kernel.hip.h),
 * 3 / 4 the one-row / tile kernel instantiated at run time with hipRTC for a shape outside kernel_dims.txt / tile_dims.txt
 * (first use costs about a second; option "no_jit" = 1 keeps such shapes on the coverage kernel) */
int tiny_batch_kernel_path(TinyBatch* b);
/* Run-time instantiated kernels and the disk cache.  With the environment variable TINYMPC_AMD_JIT_CACHE=<directory> the
 * compiled code objects are kept in that directory (one file per instantiation, keyed by the kernel sources, the compile
 * options and the hipRTC version; written atomically, so the ranks of a job may share it) and later processes load them
 * instead of compiling.  tiny_jit_compile() compiles one instantiation by its C++ name, e.g.
 * "tinympc_amd::admm_solve_kernel<5, 3, 7, false, false, 0, 0, false, 4>", without loading it: it needs no GPU, so an image
 * build can fill the directory.  tiny_jit_used() lists the names this process instantiated so far (newline separated) --
 * what to feed tiny_jit_compile() elsewhere.  Returns the code-object size in bytes (> 0; *from_disk = 1 if it was found in
 * the directory), or TINY_ERR_HIP with the reason in msg. */
long tiny_jit_compile(const char* instantiation, int* from_disk, char* msg, int msg_len);
/* The PREBUILT store (round 6): <directory of the library>/jit_prebuilt/ (TINYMPC_AMD_JIT_PREBUILT=<dir> overrides, "0": off) holds
 * code objects compiled at BUILD time -- tinympc_amd.build() calls tiny_jit_prebuild() for every name of csrc/jit_prebuilt.txt (the
 * run-time instantiated forms the BASELINE-shaped workloads of bench.py take).  They are keyed by the kernel sources, the name and the
 * options, NOT by the hipRTC version of the running process: a process that has loaded another ROCm's compiler (PyTorch wheels bring
 * their own libhiprtc / libamd_comgr) runs the build's code, and the first launch of a listed form reads a file instead of compiling.
 * Looked up after TINYMPC_AMD_JIT_CACHE, before compiling.  Returns the code-object size, 0 if the file was already there, or
 * TINY_ERR_HIP with the reason in msg.  dir NULL: the library's own store. */
long tiny_jit_prebuild(const char* instantiation, const char* dir, char* msg, int msg_len);
int tiny_jit_used(char* out, int out_len);                   /* returns the number of names; out may be NULL */
/* bytes of HBM traffic one warm solve must move per instance: 8*(nx + 8*S) + 44, S = nx*N + nu*(N-1)
 * (SURVEY.md section 8(d)); cold = 8*(nx + 2*S) + 44 */
long tiny_batch_algorithmic_bytes(TinyBatch* b, int cold);   /* cold: 0 bytes_warm, 1 bytes_cold (one_shot = 2), 2 one_shot = 1 */

/* ------------------------------------------------------------------------------------------ */
/* (C) Multi-GPU.  The path shards embarrassingly: instances are independent QPs, so a batch is
 * split over the GPUs with NO data-path collective; the only exchange is one 64-byte statistics message per shard --
 * {sum iter, sum solved, accumulated iterations, accumulated solves, the four residual maxima} -- moved by ONE RCCL
 * all-gather over xGMI and reduced (SUM / MAX) on the host.
 *
 * TinyGroup: ONE host process (the reference's callers are single-process C++ programs, examples/quadrotor_hovering.cpp)
 * drives several GPUs.  Each shard is an ordinary TinyBatch on its own device and stream (tiny_group_shard): a group
 * solve enqueues every shard without waiting.  interleaved = 0: contiguous blocks of instances per shard; 1: instance i
 * lives on shard i % n_shards (round-robin: balances batches whose iteration counts diverge).  devices == NULL: shard k
 * on GPU k % device_count; n_shards <= 0: one shard per GPU.  Shards that share a GPU (more shards than GPUs) exchange
 * through host memory -- RCCL refuses two ranks on one device -- with the identical reduction. */
typedef struct TinyGroup TinyGroup;
int tiny_group_setup(TinyGroup** out, const double* Adyn, const double* Bdyn, const double* fdyn, const double* Qdiag,
                     const double* Rdiag, double rho, int nx, int nu, int N, int batch, const int* devices, int n_shards,
                     int interleaved, int verbose);
int tiny_group_destroy(TinyGroup* g);
int tiny_group_shards(TinyGroup* g);                          /* number of shards */
TinyBatch* tiny_group_shard(TinyGroup* g, int k);             /* shard k: every tiny_batch_* call works on it */
int tiny_group_shard_indices(TinyGroup* g, int k, int* idx, int capacity);   /* caller-order instance ids of shard k; returns its size */
int tiny_group_uses_rccl(TinyGroup* g);                       /* 1: the exchange is an RCCL all-gather, 0: host memory (shared devices) */
const char* tiny_group_last_error(TinyGroup* g);
/* problem-family setters and options: forwarded to every shard (same meaning as the tiny_batch_* functions) */
int tiny_group_set_bound_constraints(TinyGroup* g, const double* x_min, const double* x_max, const double* u_min, const double* u_max);
int tiny_group_set_cone_constraints(TinyGroup* g, int n_state_cones, const int* Acx, const int* qcx, const double* cx,
                                    int n_input_cones, const int* Acu, const int* qcu, const double* cu);
int tiny_group_set_linear_constraints(TinyGroup* g, int n_state, const double* Alin_x, const double* blin_x,
                                      int n_input, const double* Alin_u, const double* blin_u);
int tiny_group_set_tv_linear_constraints(TinyGroup* g, int n_state, const double* tv_Alin_x, const double* tv_blin_x,
                                         int n_input, const double* tv_Alin_u, const double* tv_blin_u);
int tiny_group_update_settings(TinyGroup* g, double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                               int en_state_bound, int en_input_bound, int en_state_soc, int en_input_soc,
                               int en_state_linear, int en_input_linear, int en_tv_state_linear, int en_tv_input_linear);
int tiny_group_set_option(TinyGroup* g, const char* name, long value);
/* per-instance data with the FULL batch axis, host memory, the caller's instance order (flags: TINY_HOST or TINY_BROADCAST) */
int tiny_group_set(TinyGroup* g, TinyField field, const double* src, int flags);
int tiny_group_get(TinyGroup* g, TinyField field, double* dst);
int tiny_group_reset(TinyGroup* g);
/* == tiny_solve on every shard; tiny_group_solve returns 0 when every instance on every GPU converged, else 1 */
int tiny_group_solve(TinyGroup* g);
int tiny_group_solve_async(TinyGroup* g);
int tiny_group_synchronize(TinyGroup* g);
int tiny_group_get_status(TinyGroup* g, int* iter, int* solved, int* status, double* residuals);   /* caller order */
/* the one exchange: out10 = job-wide statistics in the tiny_batch_reduce_stats layout (waits for every shard) */
int tiny_group_allreduce_stats(TinyGroup* g, double* out10);
/* One process per GPU (MPI / torchrun-style hosts): the same 64-byte exchange on a communicator the CALLER created with
 * ncclCommInitRank; rccl_comm = its ncclComm_t.  Enqueued on the batch's stream behind the solve, returns when the
 * stream has drained, the same job-wide vector on every rank.  total_batch = the unsharded batch size. */
int tiny_batch_allreduce_stats(TinyBatch* b, void* rccl_comm, int n_ranks, int rank, long total_batch, double* out10);
/* Communicator plumbing for hosts that have no RCCL binding of their own: rank 0 draws the 128-byte ncclUniqueId, the host
 * hands it to every rank (MPI_Bcast, a file, torch.distributed ...), each rank joins on its device.  *comm is an ncclComm_t. */
int tiny_rccl_unique_id(void* id128);
int tiny_rccl_comm_init_rank(void** comm, int n_ranks, const void* id128, int rank, int device);
int tiny_rccl_comm_destroy(void* comm);
int tiny_rccl_comm_count(void* comm);    /* ranks of the communicator (ncclCommCount), or TINY_ERR_NULL / TINY_ERR_HIP (< 0) */
int tiny_rccl_available(void);          /* 1: librccl could be loaded next to this library's HIP runtime (no communicator is created) */
/* Only the message: the 8 doubles {sum iter, sum solved, accumulated iterations, accumulated solves, max primal_state,
 * primal_input, dual_state, dual_input} of this batch, written to device memory on the batch's stream behind the solve,
 * for hosts that run the collective themselves (bench.py hands it to torch.distributed = RCCL). */
int tiny_batch_stats_message(TinyBatch* b, void* device_out);
/* ... and the host reduction of the gathered n_shards x 8 table -> the 10-entry statistics vector (SUM over the counts, MAX over
 * the residuals; a NaN residual survives), as the two allreduce entry points run it.  Host code only: works without a GPU. */
int tiny_reduce_stats_messages(const double* table, int n_shards, long total_batch, double* out10);

/* ------------------------------------------------------------------------------------------ */
/* (B) Reference entry points over plain-data mirrors of the reference structs.
 *
 * Layout contract (x86-64, Eigen 3.4.90, measured with offsetof against the reference's own
 * src/tinympc/types.hpp, see SURVEY.md section 8(b) and tests/test_abi_layout.py):
 *   tinyMatrix  = {double* data; int64 rows; int64 cols}   (DenseStorage.h:453-457)
 *   tinyVector  = {double* data; int64 rows}                (DenseStorage.h:613-616)
 *   VectorXi    = {int* data;    int64 rows}
 * Eigen objects passed BY VALUE through C linkage arrive, under the Itanium C++ ABI, as a
 * pointer to a caller-owned temporary -- hence the `const Tiny*POD*` parameters below.          */
typedef struct { double* data; int64_t rows; int64_t cols; } TinyMatrixPOD;
typedef struct { double* data; int64_t rows; } TinyVectorPOD;
typedef struct { int* data; int64_t rows; } TinyVectorXiPOD;

typedef struct {                 /* types.hpp:32-37, 56 bytes */
    int iter;
    int solved;
    TinyMatrixPOD x;
    TinyMatrixPOD u;
} TinySolution;

typedef struct {                 /* types.hpp:43-59, 280 bytes */
    double rho;
    TinyMatrixPOD Kinf, Pinf, Quu_inv, AmBKt;
    TinyVectorPOD APf, BPf;
    TinyMatrixPOD C1, C2;
    TinyMatrixPOD dKinf_drho, dPinf_drho, dC1_drho, dC2_drho;
} TinyCache;

typedef struct {                 /* types.hpp:63-82, 88 bytes */
    double abs_pri_tol;
    double abs_dua_tol;
    int max_iter;
    int check_termination;
    int en_state_bound;
    int en_input_bound;
    int en_state_soc;
    int en_input_soc;
    int en_state_linear;
    int en_input_linear;
    int en_tv_state_linear;
    int en_tv_input_linear;
    int adaptive_rho;
    double adaptive_rho_min;
    double adaptive_rho_max;
    int adaptive_rho_enable_clipping;
} TinySettings;

typedef struct {                 /* types.hpp:88-208, 1328 bytes */
    int nx, nu, N;
    TinyMatrixPOD x, u, q, r, p, d, v, vnew, z, znew, g, y;
    TinyMatrixPOD x_min, x_max, u_min, u_max;
    int numStateCones, numInputCones;
    TinyVectorPOD cx, cu;
    TinyVectorXiPOD Acx, Acu, qcx, qcu;
    TinyMatrixPOD vc, vcnew, zc, zcnew, gc, yc;
    int numStateLinear, numInputLinear;
    TinyMatrixPOD Alin_x;
    TinyVectorPOD blin_x;
    TinyMatrixPOD Alin_u;
    TinyVectorPOD blin_u;
    TinyMatrixPOD vl, vlnew, zl, zlnew, gl, yl;
    int numtvStateLinear, numtvInputLinear;
    TinyMatrixPOD tv_Alin_x, tv_blin_x, tv_Alin_u, tv_blin_u;
    TinyMatrixPOD vl_tv, vlnew_tv, zl_tv, zlnew_tv, gl_tv, yl_tv;
    TinyVectorPOD Q, R;
    TinyMatrixPOD Adyn, Bdyn;
    TinyVectorPOD fdyn;
    TinyMatrixPOD Xref, Uref;
    TinyVectorPOD Qu;
    double primal_residual_state, primal_residual_input, dual_residual_state, dual_residual_input;
    int status;
    int iter;
} TinyWorkspace;

typedef struct {                 /* types.hpp:213-218, 32 bytes */
    TinySolution* solution;
    TinySettings* settings;
    TinyCache* cache;
    TinyWorkspace* work;
} TinySolver;

/* tiny_api.hpp:10-12 */
int tiny_setup(TinySolver** solverp, const TinyMatrixPOD* Adyn, const TinyMatrixPOD* Bdyn,
               const TinyMatrixPOD* fdyn, const TinyMatrixPOD* Q, const TinyMatrixPOD* R,
               double rho, int nx, int nu, int N, int verbose);
/* tiny_api.hpp:13-15 */
int tiny_set_bound_constraints(TinySolver* solver, const TinyMatrixPOD* x_min, const TinyMatrixPOD* x_max,
                               const TinyMatrixPOD* u_min, const TinyMatrixPOD* u_max);
/* tiny_api.hpp:16-18; positional binding of the DEFINITION (tiny_api.cpp:176-178): state first */
int tiny_set_cone_constraints(TinySolver* solver, const TinyVectorXiPOD* Acx, const TinyVectorXiPOD* qcx,
                              const TinyVectorPOD* cx, const TinyVectorXiPOD* Acu,
                              const TinyVectorXiPOD* qcu, const TinyVectorPOD* cu);
/* tiny_api.hpp:19-24 */
int tiny_set_linear_constraints(TinySolver* solver, const TinyMatrixPOD* Alin_x, const TinyVectorPOD* blin_x,
                                const TinyMatrixPOD* Alin_u, const TinyVectorPOD* blin_u);
int tiny_set_tv_linear_constraints(TinySolver* solver, const TinyMatrixPOD* tv_Alin_x, const TinyMatrixPOD* tv_blin_x,
                                   const TinyMatrixPOD* tv_Alin_u, const TinyMatrixPOD* tv_blin_u);
/* tiny_api.hpp:25-27 */
int tiny_precompute_and_set_cache(TinyCache* cache, const TinyMatrixPOD* Adyn, const TinyMatrixPOD* Bdyn,
                                  const TinyMatrixPOD* fdyn, const TinyMatrixPOD* Q, const TinyMatrixPOD* R,
                                  int nx, int nu, double rho, int verbose);
/* tiny_api.hpp:34, admm.hpp:9 */
int tiny_solve(TinySolver* solver);
int solve(TinySolver* solver);
/* tiny_api.hpp:36-43 */
int tiny_update_settings(TinySettings* settings, double abs_pri_tol, double abs_dua_tol, int max_iter,
                         int check_termination, int en_state_bound, int en_input_bound,
                         int en_state_soc, int en_input_soc, int en_state_linear, int en_input_linear,
                         int en_tv_state_linear, int en_tv_input_linear);
int tiny_set_default_settings(TinySettings* settings);
/* tiny_api.hpp:45-47 */
int tiny_set_x0(TinySolver* solver, const TinyVectorPOD* x0);
int tiny_set_x_ref(TinySolver* solver, const TinyMatrixPOD* x_ref);
int tiny_set_u_ref(TinySolver* solver, const TinyMatrixPOD* u_ref);
/* tiny_api.hpp:54: fills cache->dKinf_drho (4 x 12), dPinf_drho (12 x 12), dC1_drho (4 x 4), dC2_drho (12 x 12) with the
 * reference's quadrotor tables (tiny_api.cpp:479-540).  With settings->adaptive_rho = 1 tiny_solve then re-estimates rho
 * every 5th iteration (admm.cpp:397-423) and writes the moved cache (rho, Kinf, Pinf, C1, C2) back into the TinyCache. */
void tiny_initialize_sensitivity_matrices(TinySolver* solver);
/* codegen.hpp:9-18.  The reference freezes one TinySolver into Eigen-initialised C++ for microcontrollers; these freeze the
 * solver's PROBLEM FAMILY (dimensions, settings, the cache as it stands, dynamics, costs, bounds, cones, half-spaces,
 * references; with adaptive_rho on also the sensitivity tables) into a plain-C project for this library:
 * <dir>/tinympc/tiny_data.h, <dir>/src/tiny_data.c (round-trip literals + tiny_generated_batch / tiny_generated_group, which
 * rebuild the device-resident batch -- or the multi-GPU group -- of any size from the data), <dir>/src/tiny_main.c, <dir>/Makefile. */
int tiny_codegen(TinySolver* solver, const char* output_dir, int verbose);
int tiny_codegen_with_sensitivity(TinySolver* solver, const char* output_dir, TinyMatrixPOD* dK, TinyMatrixPOD* dP,
                                  TinyMatrixPOD* dC1, TinyMatrixPOD* dC2, int verbose);
int codegen_create_directories(const char* output_dir, int verbose);
int codegen_data_header(const char* output_dir, int verbose);
int codegen_data_source(TinySolver* solver, const char* output_dir, int verbose);
int codegen_example(const char* output_dir, int verbose);
/* admm.hpp:12-17: the individual phases of one ADMM iteration on a solver's workspace.  Each uploads the workspace,
 * runs the phase on the GPU (tiny_batch_phase with a batch of one) and writes the fields the reference function
 * writes back into the workspace. */
void update_linear_cost(TinySolver* solver);
void backward_pass_grad(TinySolver* solver);
void forward_pass(TinySolver* solver);
void update_slack(TinySolver* solver);
void update_dual(TinySolver* solver);
bool termination_condition(TinySolver* solver);
/* admm.hpp:25, :34 -- `tinyVector project_soc(tinyVector s, float mu)` and
 * `tinyVector project_hyperplane(const tinyVector& z, const tinyVector& a, tinytype b)`: under the Itanium ABI the
 * Eigen return value is constructed by the callee in caller-provided storage (hidden first argument, returned in
 * rax) and the by-value Eigen argument arrives as a pointer; the result buffer is malloc'd (Eigen frees it). */
TinyVectorPOD* project_soc(TinyVectorPOD* result, const TinyVectorPOD* s, float mu);
TinyVectorPOD* project_hyperplane(TinyVectorPOD* result, const TinyVectorPOD* z, const TinyVectorPOD* a, double b);
/* rho_benchmark.hpp:5-94 -- the adaptive-rho module's own helpers.  Upstream they are C++ functions (no extern "C"), so the
 * library exports them under their Itanium names and a C header cannot declare them; for C++ callers the reference's
 * rho_benchmark.hpp IS the declaration.  The structs they take, as plain data (layout checked against the real header):
 *   void     initialize_format_matrices(RhoAdapter*, int nx, int nu, int N)                    rho_benchmark.cpp:14-43
 *   void     format_matrices(RhoAdapter*, x, u, v, z, g, y (const tinyMatrix&), TinyCache*, TinyWorkspace*, int N)   :45-145
 *   void     compute_residuals(RhoAdapter*, tinytype* pri_res, dual_res, pri_norm, dual_norm)  :147-178   (on the GPU)
 *   tinytype predict_rho(RhoAdapter*, pri_res, dual_res, pri_norm, dual_norm, current_rho)     :180-201
 *   void     update_matrices_with_derivatives(TinyCache*, tinytype new_rho)                    :203-217
 *   void     benchmark_rho_adaptation(RhoAdapter*, x, u, v, z, g, y, TinyCache*, TinyWorkspace*, int N, RhoBenchmarkResult*)   :219-253
 *   uint32_t micros()                                                                          :9-11 */
typedef struct {
    double rho_min, rho_max;
    bool clip, matrices_initialized;
    TinyMatrixPOD A_matrix, z_vector, y_vector, x_decision, P_matrix, q_vector;
    TinyMatrixPOD Ax_vector, r_prim_vector, r_dual_vector, Px_vector, ATy_vector;
    int format_nx, format_nu, format_N;
} TinyRhoAdapterPOD;
typedef struct {
    uint32_t time_us;
    double initial_rho, final_rho, pri_res, dual_res, pri_norm, dual_norm;
} TinyRhoBenchmarkResultPOD;
/* NEW: n solvers that share one cache / settings / bounds, solved in ONE launch (gather from and
 * scatter to ordinary TinySolver workspaces).  Returns 0 when all converged, else 1. */
int tiny_solve_batch(TinySolver** solvers, int n);
/* NEW: the reference has no destroy (tiny_api.cpp:25-29 leaks); this frees host + device state. */
int tiny_destroy(TinySolver* solver);

#ifdef __cplusplus
}
#endif
#endif /* TINYMPC_AMD_H */
