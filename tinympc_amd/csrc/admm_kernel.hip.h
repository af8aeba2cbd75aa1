// admm_kernel.hip.h -- the MI355X (gfx950 / CDNA4) ADMM solve kernel.
//
// What it computes: one tiny_solve() (reference src/tinympc/admm.cpp:331-455) for each of
// `batch` independent MPC QPs that share one TinyCache (Kinf, Pinf, Quu_inv, AmBKt, APf, BPf)
// and one set of bounds / cones / settings.  Per ADMM iteration, in the reference's order:
//   update_linear_cost (admm.cpp:262-297) -> backward_pass_grad (:13-20) -> forward_pass (:25-32)
//   -> update_slack (:81-135, box + second-order cone) -> update_dual (:219-235)
//   -> termination_condition (:310-328) -> v = vnew, z = znew (:445-446).
//
// Mapping (designed for CDNA4, not translated from anything):
//   * A 16-lane DPP row of a wave64 owns ONE problem instance; a wavefront carries 4.
//     Lane j of the row owns ROW j of the stacked knot vector [x_i ; u_i] (j < nx: state row j,
//     nx <= j < nx+nu: input row j-nx) for EVERY knot point i.  All per-element ADMM state
//     (x/u, vnew/znew, v/z, g/y, the reference-cost term) therefore lives in that lane's
//     registers as N-long arrays, and every element-wise phase (linear cost, slack projection,
//     dual update, residuals) is lane-local: no LDS, no shuffles.
//   * The Riccati sweeps need y = M * w with M a shared <=16x16 matrix and w spread one entry
//     per lane.  Lane j keeps row j of M in registers and the row-broadcast of w_k comes for free
//     from the DPP `row_newbcast:k` modifier folded into the FMA:
//         v_fmac_f64_dpp acc, w, M[j][k] row_newbcast:k       (gfx90a+ "DP-ALU DPP")
//     i.e. one FP64 FMA issue slot per matrix column, no LDS traffic, no cross-lane reduction.
//     (A broadcast-through-LDS formulation is LDS-bandwidth bound at 2x the FMA time on CDNA4:
//     8 B of LDS read per FMA vs 256 B/clk/CU LDS and 64 FP64 FMA/clk/CU.)
//   * No MFMA: FP64 MFMA on MI355X runs at the vector FP64 rate and the path is bounded by HBM
//     (few-iteration warm solves) or FP64 VALU issue (cold solves), never by matrix throughput.
//   * HBM layout = "knot-point interleaved" records, [instance][knot i][row j] with row stride
//     nx+nu, so a DPP row reads/writes one contiguous (nx+nu)*8-byte segment per knot point
//     (128 B for the quadrotor) and consecutive instances are contiguous: every byte of every
//     touched cache line is used.
//   * One workgroup = one wavefront (no __syncthreads anywhere) = by default one tile of 4 instances, so that the hardware
//     dispatcher balances tiles with different iteration counts (with a capped grid the waves walk the tiles with a grid
//     stride instead).  Rows that converge early are masked off by EXEC (per-row exit), the wave leaves the iteration loop
//     when its last row has converged.
//   * Template variants: SOC (second-order-cone slacks), DBG (keep q, r, p, d), MODE (how the row broadcast is issued),
//     LIN / KMAX (static / time-varying half-spaces), HET (per-instance problem data).  kernel_dims.txt lists the shapes
//     compiled into the library; every other shape / variant is instantiated at run time from this very header (jit.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tinympc_amd {

// ---- table layout shared with the host (batch_api.cpp builds it) --------------------------
// All lane tables are [column k][lane j] with 16 lanes, doubles.
enum : int {
    TAB_MB = 0,          // backward:  cols k<nx multiply p_{i+1}[k], cols nx.. multiply r_i[k-nx]
    TAB_MF1 = 256,       // forward 1: cols k<nx multiply x_i[k]   (state lanes: A, input lanes: -Kinf)
    TAB_MF2 = 512,       // forward 2: cols nx.. multiply u_i[k-nx] (state lanes: B)
    TAB_PT = 768,        // terminal:  cols k<nx multiply Xref[k, N-1] (state lanes: Pinf[k][j])
    TAB_VEC = 1024,      // 16-entry lane vectors, see VEC_* below
    TAB_BOUNDS = 1024 + 16 * 16,
};
enum : int {
    VEC_CB = 0,          // backward constant: APf (state lanes), Quu_inv*BPf (input lanes)
    VEC_CF = 1,          // forward constant: fdyn (state lanes)
    VEC_QR = 2,          // work->Q / work->R (user diagonal + rho)
    VEC_SMASK = 3,       // 1.0 on state lanes else 0
    VEC_NIM = 4,         // -1.0 on input lanes else 0
    VEC_SOCFLAG = 5,     // 1.0 when this lane's cone slack is enabled (admm.cpp:102-109)
    VEC_CONE_BASE = 6,   // first lane of the cone this lane belongs to, or -1
    VEC_CONE_MU = 7,     // cone coefficient (double; truncated to float as admm.cpp:39 does)
    VEC_LINFLAG = 9,     // 1.0 when this lane has a static linear-constraint slack (admm.cpp:138-145); 8 = VEC_RHO
    VEC_TLINFLAG = 10,   // same for the time-varying family (admm.cpp:176-183)
    VEC_COUNT = 16,
};
// Half-space tables of the register-resident linear-constraint variant, appended after the bounds.  Each entry is
// [constraint k < LIN_KMAX][16 lanes]: the coefficient of this lane's row (state lanes: Alin_x[k][j], input lanes:
// Alin_u[k][j-nx]), the offset b_k and ||a_k||^2 of this lane's family (b = +inf where there is no constraint k).
// The time-varying tables carry one such block per slot (input lanes shifted by one knot, like the bounds).
enum : int { LIN_KMAX = 4, LIN_KMAX_BIG = 32 };      // half-spaces per knot and family: compiled-in variants / largest run-time instantiated one
// Lane tables of the adaptive-rho variant (ADAPT; SolveArgs::atab), [column k][16 lanes] each:
enum : int {
    ATAB_AT = 0,         // cols k<nx multiply g_{i+1}[k]: state lanes A[k][j] (A' g), input lanes B[k][j-nx] (B' g)
    ATAB_DK = 256,       // dKinf_drho: state lanes cols k<nu dK[k][j] (their -Kinf' entries), input lanes cols k<nx dK[j-nx][k]
    ATAB_DP = 512,       // dPinf_drho: state lanes cols k<nx dP[k][j] (column j, the one lane j keeps up to date)
    ATAB_DC1 = 768,      // dC1_drho:   lanes j<nu cols k<nu dC1[k][j]
    ATAB_DC2 = 1024,     // dC2_drho:   state lanes cols k<nx dC2[k][j]
    ATAB_DOUBLES = 1280,
};
static inline int tab_lin_offset(int N) { return TAB_BOUNDS + 2 * N * 16; }
static inline int tab_tlin_offset(int N, int kmax = LIN_KMAX) { return tab_lin_offset(N) + 3 * kmax * 16; }
static inline int tab_doubles(int N, int kmax = LIN_KMAX) { return tab_tlin_offset(N, kmax) + 3 * N * kmax * 16; }

struct SolveArgs {
    const double* tab;        // tab_doubles(N) doubles
    const double* x0;         // [batch][nx]
    const double* ref;        // KPI: Xref/Uref
    double* prim;             // KPI: x/u        (out; also in when a cone is enabled, admm.cpp:352-357)
    double* slack;            // KPI: vnew/znew  (in/out)
    double* dual;             // KPI: g/y        (in/out)
    double* slack_prev;       // KPI: v/z        (in/out)
    double* cslack;           // KPI: vcnew/zcnew (out, SOC only)
    double* cdual;            // KPI: gc/yc      (in/out, SOC only)
    int4* status;             // [batch] {iter, solved, status (1 | 11), checked}
    double* resid;            // [batch][4] {pri_state, pri_input, dua_state, dua_input}
    double* x0_next;          // optional [batch][nx]: plant step x1 = A x0 + B u0 + f (may alias x0)
    double* dbg_qr;           // optional KPI: q/r of the last iteration (work->q, work->r)
    double* dbg_pd;           // optional KPI: p/d of the last iteration (work->p, work->d)
    uint2* accum;             // optional [batch] {iterations, solves converged} accumulated over successive solves
    int* iter_log;            // optional [steps][batch]: per-MPC-step iteration count (negative: not converged), steps > 1
    double* u0_log;           // optional [steps][batch][nu]: the applied control u[:,0] of every fused MPC step
    double rho, tol_pri, tol_dua;
    int batch, max_iter, check_termination;   // max_iter: where THIS launch stops (a stage of a split solve: its cap) -- the last termination test
                                              // before it is always a full one, so the residuals a launch leaves are those of its last check
    int steps;                // closed-loop MPC steps fused into this launch (>= 1; > 1 implies the plant step)
    // Reference-trajectory window (examples/quadrotor_tracking.cpp:65,89): when traj != nullptr the state
    // reference of MPC step k is traj[k + offset_b + 0 .. N-1][nx] (shared by all instances) instead of the
    // Xref part of the ref record; the window advances one knot per fused step.
    const double* traj;       // [traj_points][nx] or nullptr
    const int* traj_offsets;  // optional [batch] per-instance start offsets
    int traj_points, traj_step0;
    int reset_duals;          // 1: g = 0, y = 0 before every solve (examples/quadrotor_tracking.cpp:92-93)
    // Heterogeneous problem families (riccati_kernel.hip.h): per-instance matrix/vector tables
    // ([batch][TAB_BOUNDS] doubles, same layout as `tab`); bounds, cones and masks stay shared.
    const double* het_tabs;
    // register-resident linear constraints (LIN variants): KPI records of vlnew|zlnew, gl|yl, vlnew_tv|zlnew_tv,
    // gl_tv|yl_tv and the number of half-spaces applied per knot (max over the state / input families)
    double *lslack, *ldual, *tlslack, *tldual;
    int n_lin, n_tlin;
    // One-shot solves (SURVEY.md 8(d) bytes_cold): 1 = the warm-start state is taken as zero (the state after
    // tiny_setup / a reset) without being read.  store_mask: which records the launch writes back -- bit 0 x|u,
    // bit 1 vnew|znew (= solution->x|u), bit 2 g|y, bit 3 v|z, bit 4 the cone / linear slack and dual records, bit 5 (without bit 0) only the first
    // knot of x|u.
    int cold, store_mask;
    // Resumed solves (batch_dispatch.hip "repack_after"): an earlier launch capped at iter_base iterations has stored the ADMM
    // state of the instances that did not converge; index[0 .. *count) lists them and this launch carries on from
    // iteration iter_base with four of them per wave again.  The cone / half-space slacks are then read from their own records
    // instead of being initialised from x (admm.cpp:352-374 runs once per solve).  index == nullptr: a plain launch.
    // next_index != nullptr: this launch is itself capped below the solver's max_iter, and lists the instances it leaves
    // open for the next stage (one atomic per wave that has any; the order in the list is irrelevant to the results).
    const int* index;
    const int* count;
    int iter_base;
    int* next_index;
    int* next_count;
    // Adaptive rho (ADAPT variants; admm.cpp:397-423 + rho_benchmark.cpp): the cache is per-instance STATE -- rho, Kinf, Pinf
    // (and the dead copies C1, C2) move every 5th iteration and persist from solve to solve.  arho [batch]; aK [batch][nu*nx],
    // aP [batch][nx*nx], aC1 [batch][nu*nu], aC2 [batch][nx*nx] column-major (aC1 / aC2 may be null); atab: ATAB_* lane tables.
    double *arho, *aK, *aP, *aC1, *aC2;
    const double* atab;
    double arho_min, arho_max;
    int aclip;
    // 1: every instance has the same Xref | Uref record (set with TINY_BROADCAST, or never set): all rows read instance 0's
    // record -- one L2-resident line set instead of 8S bytes of HBM per instance
    int ref_shared;
    // tile kernel, dynamic form (tile_kernel.hip.h DYN): ONE device-wide counter of the instances handed out so far, zeroed by
    // the host before the launch; a slot of a persistent wave takes the next instance off it the moment it is free.
    // one-row kernel: the counter of 4-instance tiles handed out BEYOND the grid's first ones (null: fixed grid stride)
    int* work_counter;
    // one-row kernel, plain launches: 1 = the grid walks the tiles from the LAST one down.  Successive warm launches over one batch
    // alternate the direction, so that a launch begins with the records its predecessor touched last -- the ones the 256 MiB
    // Infinity Cache in front of HBM still holds (a batch whose records exceed it would otherwise stream through it without a hit).
    // Instances are independent: the order changes nothing in the results.
    int reverse;
    // one-row kernel, plain launches: slot s of the grid solves instance perm[s] (null: instance s).  A fused closed-loop launch that
    // is cut into stretches of MPC steps (batch_dispatch.hip "step_regroup") hands every stretch the instances ordered by the iteration
    // count of their last solve: the four rows of a wave run in lock step, so a wave costs what its slowest row costs, and rows
    // that took alike counts at the last step take alike counts at the next ones.  Same reason as above: nothing in the results.
    const int* perm;
    int perm_count;           // slots of `perm` (a launch may take a part of the batch: batch_dispatch.hip runs two halves on two streams)
    // tile kernel, forms that stream v|z out of the register file (TILE_LM_VPG): where the stream goes.  null: the instance's v|z record
    // (what a warm-started solve must leave there anyway); a ONE-SHOT launch promises not to touch that record, and hands a scratch
    // array of the same shape here instead (round 6: one_shot rides on the shape's fast box form, not on the all-in-registers one)
    double* vz_stream;
    // one-row kernel, PREFETCH form (template PF; round 6): persistent waves whose NEXT tile's records are on their way into the wave's
    // LDS buffer (LDS-DMA, global_load_lds_dwordx4: no register holds them) while the current tile iterates -- at two waves per SIMD
    // (247 VGPRs) a wave's load phase otherwise hides behind ONE neighbour only (tools/ubench/ubench_stream_forms.hip: the record traffic
    // of a warm solve beyond the Infinity Cache 4.3 -> 5.4 TB/s).  pf_mask: which record arrays travel that way (bit 0 vnew|znew, 1 g|y,
    // 2 v|z, 3 Xref|Uref; the others are read as in the plain form); the buffer is the launch's dynamic LDS, [x0 piece][arrays in bit
    // order] in 1-KiB pieces.  Which tiles a wave takes: its first pf_static ones by grid stride (block index + i * grid: no
    // communication, and no value in flight that the wave would have to wait for at a tile's top -- the stores of the tile before stay
    // in flight), the rest by TICKET, so that the waves drain together whatever their tiles cost.  pf_counter: ticket counters, one per
    // SHARD (pf_shards = 8 or 1; 64 bytes apart): ticket k of shard x is tile pf_static * grid + k * pf_shards + x, a wave belongs to
    // shard blockIdx % pf_shards (one counter for the whole device saturates at ~90 returning atomics per microsecond: 65 536 tiles
    // would take 0.7 ms).  The counters are never reset: a shard's waves draw exactly (its ticketed tiles) + (its waves) tickets per
    // launch, pf_base[x] is where this launch's begin (the host keeps the sums)
    int pf_mask, pf_shards, pf_static;
    unsigned* pf_counter;
    unsigned pf_base[8];
};

// ---- DPP row-broadcast FMA blocks ------------------------------------------------------------
// One asm statement per block so that the two wait states a DPP read needs after a VALU write of
// its source (CDNA ISA "VALU writes VGPR -> DPP reads that VGPR") are paid once (the leading
// s_nop 1), and no compiler-inserted copy can land between the FMAs.  The accumulators are
// EARLY-CLOBBER ("&"): they are written while src / m[] are still being read, so they must never
// share a register with an input (without it hipcc happily aliases an accumulator with src).
//
//   TF_  : v_fmac_f64_dpp acc, src, m[k] row_newbcast:COL0+k      acc += bcast(src, lane COL0+k) * m[k]
#define TF_(acc, mi, k) "v_fmac_f64_dpp %" #acc ", %2, %" #mi " row_newbcast:%3+" #k " row_mask:0xf bank_mask:0xf\n\t"
// two-accumulator chains, every column an fmac (both accumulators carry a value in)
#define TB1 TF_(0, 4, 0)
#define TB2 TB1 TF_(1, 5, 1)
#define TB3 TB2 TF_(0, 6, 2)
#define TB4 TB3 TF_(1, 7, 3)
#define TB5 TB4 TF_(0, 8, 4)
#define TB6 TB5 TF_(1, 9, 5)
#define TB7 TB6 TF_(0, 10, 6)
#define TB8 TB7 TF_(1, 11, 7)
#define TB9 TB8 TF_(0, 12, 8)
#define TB10 TB9 TF_(1, 13, 9)
#define TB11 TB10 TF_(0, 14, 10)
#define TB12 TB11 TF_(1, 15, 11)
#define TB13 TB12 TF_(0, 16, 12)
#define TB14 TB13 TF_(1, 17, 13)
#define TB15 TB14 TF_(0, 18, 14)
#define TB16 TB15 TF_(1, 19, 15)
// single-accumulator chain (operand numbering: %0 acc, %1 src, %2 COL0, %3.. m[k])
#define TS_(mi, k) "v_fmac_f64_dpp %0, %1, %" #mi " row_newbcast:%2+" #k " row_mask:0xf bank_mask:0xf\n\t"
#define TS1 TS_(3, 0)
#define TS2 TS1 TS_(4, 1)
#define TS3 TS2 TS_(5, 2)
#define TS4 TS3 TS_(6, 3)
#define TS5 TS4 TS_(7, 4)
#define TS6 TS5 TS_(8, 5)
#define TS7 TS6 TS_(9, 6)
#define TS8 TS7 TS_(10, 7)
#define TS9 TS8 TS_(11, 8)
#define TS10 TS9 TS_(12, 9)
#define TS11 TS10 TS_(13, 10)
#define TS12 TS11 TS_(14, 11)
#define TS13 TS12 TS_(15, 12)
#define TS14 TS13 TS_(16, 13)
#define TS15 TS14 TS_(17, 14)
#define TS16 TS15 TS_(18, 15)
#define TM1 "v"(m[0])
#define TM2 TM1, "v"(m[1])
#define TM3 TM2, "v"(m[2])
#define TM4 TM3, "v"(m[3])
#define TM5 TM4, "v"(m[4])
#define TM6 TM5, "v"(m[5])
#define TM7 TM6, "v"(m[6])
#define TM8 TM7, "v"(m[7])
#define TM9 TM8, "v"(m[8])
#define TM10 TM9, "v"(m[9])
#define TM11 TM10, "v"(m[10])
#define TM12 TM11, "v"(m[11])
#define TM13 TM12, "v"(m[12])
#define TM14 TM13, "v"(m[13])
#define TM15 TM14, "v"(m[14])
#define TM16 TM15, "v"(m[15])
#define RING_CASE(K)                                                                        \
    if constexpr (NCOL == K) {                                                              \
        asm("s_nop 1\n\t" TB##K : "+&v"(a0), "+&v"(a1) : "v"(src), "i"(COL0), TM##K);        \
    }
#define RING1_CASE(K)                                                                       \
    if constexpr (NCOL == K) {                                                              \
        asm("s_nop 1\n\t" TS##K : "+&v"(a0) : "v"(src), "i"(COL0), TM##K);                   \
    }

template <int C>
__device__ __forceinline__ double row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + C, 0xf, 0xf, false);   // DPP_ROW_NEWBCAST0 + C
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + C, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int COL0, int NCOL, int K>
struct RingB {
    static __device__ __forceinline__ void run(double& a0, double& a1, double src, const double* m) {
        if constexpr (K < NCOL) {
            if constexpr ((K & 1) == 0) a0 = fma(row_bcast<COL0 + K>(src), m[K], a0);
            else a1 = fma(row_bcast<COL0 + K>(src), m[K], a1);
            RingB<COL0, NCOL, K + 1>::run(a0, a1, src, m);
        }
    }
};

// a{0,1} += sum_{k<NCOL} bcast(src, lane COL0+k) * m[k]   (even k -> a0, odd k -> a1)
// MODE 0: fused v_fmac_f64_dpp (1 issue slot / column); MODE 1: v_mov_b32_dpp x2 + v_fma_f64.
template <int MODE, int COL0, int NCOL>
__device__ __forceinline__ void ring(double& a0, double& a1, double src, const double* m) {
    static_assert(NCOL >= 1 && NCOL <= 16 && COL0 + NCOL <= 16, "one DPP row");
    if constexpr (MODE == 0) {
        RING_CASE(1) RING_CASE(2) RING_CASE(3) RING_CASE(4) RING_CASE(5) RING_CASE(6) RING_CASE(7) RING_CASE(8)
        RING_CASE(9) RING_CASE(10) RING_CASE(11) RING_CASE(12) RING_CASE(13) RING_CASE(14) RING_CASE(15) RING_CASE(16)
    } else {
        RingB<COL0, NCOL, 0>::run(a0, a1, src, m);
    }
}
// single accumulator chain: a0 += sum_k bcast(src, COL0+k) * m[k]
template <int COL0, int NCOL>
__device__ __forceinline__ void ring1(double& a0, double src, const double* m) {
    static_assert(NCOL >= 1 && NCOL <= 16 && COL0 + NCOL <= 16, "one DPP row");
    RING1_CASE(1) RING1_CASE(2) RING1_CASE(3) RING1_CASE(4) RING1_CASE(5) RING1_CASE(6) RING1_CASE(7) RING1_CASE(8)
    RING1_CASE(9) RING1_CASE(10) RING1_CASE(11) RING1_CASE(12) RING1_CASE(13) RING1_CASE(14) RING1_CASE(15) RING1_CASE(16)
}
// init + sum_{k<NA} bcast(srcA, k) mA[k] + sum_{k<NB} bcast(srcB, NA+k) mB[k]
//   MODE 0: two accumulator chains (fused DPP FMA);  MODE 2: one chain;  MODE 1: compiler-scheduled mov_dpp + fma
template <int MODE, int NA, int NB>
__device__ __forceinline__ double ring_sum2(double init, double srcA, const double* mA, double srcB, const double* mB) {
    if constexpr (MODE == 2) {
        double a0 = init;
        ring1<0, NA>(a0, srcA, mA);
        ring1<NA, NB>(a0, srcB, mB);
        return a0;
    } else {
        double a0 = init, a1 = 0.0;
        ring<MODE, 0, NA>(a0, a1, srcA, mA);
        ring<MODE, NA, NB>(a0, a1, srcB, mB);
        return a0 + a1;
    }
}
template <int MODE, int COL0, int NCOL>
__device__ __forceinline__ double ring_sum(double init, double src, const double* m) {
    if constexpr (MODE == 2) {
        double a0 = init;
        ring1<COL0, NCOL>(a0, src, m);
        return a0;
    } else {
        double a0 = init, a1 = 0.0;
        ring<MODE, COL0, NCOL>(a0, a1, src, m);
        return a0 + a1;
    }
}
// short blocks (the forward pass' B u_i): always one chain when fused
template <int MODE, int COL0, int NCOL>
__device__ __forceinline__ double ring_short(double init, double src, const double* m) {
    if constexpr (MODE == 1) return ring_sum<1, COL0, NCOL>(init, src, m);
    else { double a0 = init; ring1<COL0, NCOL>(a0, src, m); return a0; }
}


// HALF rows: lanes 0-7 and 8-15 of a DPP row carry two different instances.  `row_newbcast:k` broadcasts ONE lane to the whole row, so
// a column of the mat-vec takes two instructions, each writing one half only (bank_mask: write-enable per bank of 4 lanes; honoured by
// the DP-ALU DPP form at no cost, tools/ubench/ubench_dpp_bankmask.hip): lanes 0-7 get lane k, lanes 8-15 get lane 8+k.  Per instance
// that is the same number of FMA issue slots as a full row gives -- the gain is everywhere else: every lane-local instruction (the
// slot update is 13 of the 25 instructions per slot at (4,2)) now serves twice the instances, and so does every register.
// HARDWARE NOTE (measured, tools/ubench/ubench_dpp_bankmask2.hip; not in the LLVM hazard tables): a bank-masked DPP instruction
// writes its DISABLED lanes back with the value of vdst it read at operand fetch ("old"), and that read is NOT interlocked against a
// VALU write of vdst in the instruction before -- `fmac bank_mask:0x3` directly followed by `fmac bank_mask:0xc` on the same
// accumulator loses the first one's result (the second writes the stale lanes 0-7 back).  One wait state in between is enough.
// So a chain runs all its low-half FMAs (their enabled lanes accumulate through the interlocked src2 path; the disabled lanes keep
// being re-written with a value that does not change), then `s_nop 0`, then all its high-half FMAs.
#define THL_(mi, k) "v_fmac_f64_dpp %0, %1, %" #mi " row_newbcast:%2+" #k " row_mask:0xf bank_mask:0x3\n\t"
#define THH_(mi, k) "v_fmac_f64_dpp %0, %1, %" #mi " row_newbcast:8+%2+" #k " row_mask:0xf bank_mask:0xc\n\t"
#define THL1 THL_(3, 0)
#define THL2 THL1 THL_(4, 1)
#define THL3 THL2 THL_(5, 2)
#define THL4 THL3 THL_(6, 3)
#define THL5 THL4 THL_(7, 4)
#define THL6 THL5 THL_(8, 5)
#define THL7 THL6 THL_(9, 6)
#define THL8 THL7 THL_(10, 7)
#define THH1 THH_(3, 0)
#define THH2 THH1 THH_(4, 1)
#define THH3 THH2 THH_(5, 2)
#define THH4 THH3 THH_(6, 3)
#define THH5 THH4 THH_(7, 4)
#define THH6 THH5 THH_(8, 5)
#define THH7 THH6 THH_(9, 6)
#define THH8 THH7 THH_(10, 7)
#define RINGH_CASE(K)                                                                       \
    if constexpr (NCOL == K) {                                                              \
        asm("s_nop 1\n\t" THL##K "s_nop 0\n\t" THH##K : "+&v"(a0) : "v"(src), "i"(COL0), TM##K);   \
    }
// a0 += sum_k bcast_half(src, COL0+k) * m[k]   (each half of the row reads its OWN lanes COL0+k)
template <int COL0, int NCOL>
__device__ __forceinline__ void ring1_half(double& a0, double src, const double* m) {
    static_assert(NCOL >= 1 && COL0 + NCOL <= 8, "one half row");
    RINGH_CASE(1) RINGH_CASE(2) RINGH_CASE(3) RINGH_CASE(4) RINGH_CASE(5) RINGH_CASE(6) RINGH_CASE(7) RINGH_CASE(8)
}


// ---- fused sweep steps (single-chain mode, box-only variants) --------------------------------------------------------------
// The leading `s_nop 1` of a DPP block only waits out the two wait states a DPP read needs after a VALU write of its source.
// Here the lane-local instructions a sweep step needs anyway stand in front of the DPP chain INSIDE the same asm statement --
// the linear-cost terms of the backward step, the first half of the slot update of the forward step -- so the wait states are
// filled with work (38 s_nop per (12,4,10) iteration gone, +2.7 % measured as an upper bound with the nops simply deleted)
// and no compiler-inserted copy can land between the producer of a source and its first DPP read.  Same instructions, same
// operand order as the unfused code: bit-identical results.
#define FCA1 "v_fmac_f64_dpp %[acc], %[sa], %[a0] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
#define FCA2 FCA1 "v_fmac_f64_dpp %[acc], %[sa], %[a1] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
#define FCA3 FCA2 "v_fmac_f64_dpp %[acc], %[sa], %[a2] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
#define FCA4 FCA3 "v_fmac_f64_dpp %[acc], %[sa], %[a3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define FCA5 FCA4 "v_fmac_f64_dpp %[acc], %[sa], %[a4] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
#define FCA6 FCA5 "v_fmac_f64_dpp %[acc], %[sa], %[a5] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
#define FCA7 FCA6 "v_fmac_f64_dpp %[acc], %[sa], %[a6] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
#define FCA8 FCA7 "v_fmac_f64_dpp %[acc], %[sa], %[a7] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
#define FCA9 FCA8 "v_fmac_f64_dpp %[acc], %[sa], %[a8] row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
#define FCA10 FCA9 "v_fmac_f64_dpp %[acc], %[sa], %[a9] row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
#define FCA11 FCA10 "v_fmac_f64_dpp %[acc], %[sa], %[a10] row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
#define FCA12 FCA11 "v_fmac_f64_dpp %[acc], %[sa], %[a11] row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
#define FCA13 FCA12 "v_fmac_f64_dpp %[acc], %[sa], %[a12] row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
#define FCA14 FCA13 "v_fmac_f64_dpp %[acc], %[sa], %[a13] row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
#define FCA15 FCA14 "v_fmac_f64_dpp %[acc], %[sa], %[a14] row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
#define FCA16 FCA15 "v_fmac_f64_dpp %[acc], %[sa], %[a15] row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
#define FCB1 "v_fmac_f64_dpp %[acc], %[sb], %[b0] row_newbcast:%[c0]+0 row_mask:0xf bank_mask:0xf\n\t"
#define FCB2 FCB1 "v_fmac_f64_dpp %[acc], %[sb], %[b1] row_newbcast:%[c0]+1 row_mask:0xf bank_mask:0xf\n\t"
#define FCB3 FCB2 "v_fmac_f64_dpp %[acc], %[sb], %[b2] row_newbcast:%[c0]+2 row_mask:0xf bank_mask:0xf\n\t"
#define FCB4 FCB3 "v_fmac_f64_dpp %[acc], %[sb], %[b3] row_newbcast:%[c0]+3 row_mask:0xf bank_mask:0xf\n\t"
#define FCB5 FCB4 "v_fmac_f64_dpp %[acc], %[sb], %[b4] row_newbcast:%[c0]+4 row_mask:0xf bank_mask:0xf\n\t"
#define FCB6 FCB5 "v_fmac_f64_dpp %[acc], %[sb], %[b5] row_newbcast:%[c0]+5 row_mask:0xf bank_mask:0xf\n\t"
#define FCB7 FCB6 "v_fmac_f64_dpp %[acc], %[sb], %[b6] row_newbcast:%[c0]+6 row_mask:0xf bank_mask:0xf\n\t"
#define FCB8 FCB7 "v_fmac_f64_dpp %[acc], %[sb], %[b7] row_newbcast:%[c0]+7 row_mask:0xf bank_mask:0xf\n\t"
#define FCB9 FCB8 "v_fmac_f64_dpp %[acc], %[sb], %[b8] row_newbcast:%[c0]+8 row_mask:0xf bank_mask:0xf\n\t"
#define FCB10 FCB9 "v_fmac_f64_dpp %[acc], %[sb], %[b9] row_newbcast:%[c0]+9 row_mask:0xf bank_mask:0xf\n\t"
#define FCB11 FCB10 "v_fmac_f64_dpp %[acc], %[sb], %[b10] row_newbcast:%[c0]+10 row_mask:0xf bank_mask:0xf\n\t"
#define FCB12 FCB11 "v_fmac_f64_dpp %[acc], %[sb], %[b11] row_newbcast:%[c0]+11 row_mask:0xf bank_mask:0xf\n\t"
#define FCB13 FCB12 "v_fmac_f64_dpp %[acc], %[sb], %[b12] row_newbcast:%[c0]+12 row_mask:0xf bank_mask:0xf\n\t"
#define FCB14 FCB13 "v_fmac_f64_dpp %[acc], %[sb], %[b13] row_newbcast:%[c0]+13 row_mask:0xf bank_mask:0xf\n\t"
#define FCB15 FCB14 "v_fmac_f64_dpp %[acc], %[sb], %[b14] row_newbcast:%[c0]+14 row_mask:0xf bank_mask:0xf\n\t"
#define FCB16 FCB15 "v_fmac_f64_dpp %[acc], %[sb], %[b15] row_newbcast:%[c0]+15 row_mask:0xf bank_mask:0xf\n\t"
#define FFA1 "v_fmac_f64_dpp %[t], %[xi], %[a0] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
#define FFA2 FFA1 "v_fmac_f64_dpp %[t], %[xi], %[a1] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
#define FFA3 FFA2 "v_fmac_f64_dpp %[t], %[xi], %[a2] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
#define FFA4 FFA3 "v_fmac_f64_dpp %[t], %[xi], %[a3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define FFA5 FFA4 "v_fmac_f64_dpp %[t], %[xi], %[a4] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
#define FFA6 FFA5 "v_fmac_f64_dpp %[t], %[xi], %[a5] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
#define FFA7 FFA6 "v_fmac_f64_dpp %[t], %[xi], %[a6] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
#define FFA8 FFA7 "v_fmac_f64_dpp %[t], %[xi], %[a7] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
#define FFA9 FFA8 "v_fmac_f64_dpp %[t], %[xi], %[a8] row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
#define FFA10 FFA9 "v_fmac_f64_dpp %[t], %[xi], %[a9] row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
#define FFA11 FFA10 "v_fmac_f64_dpp %[t], %[xi], %[a10] row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
#define FFA12 FFA11 "v_fmac_f64_dpp %[t], %[xi], %[a11] row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
#define FFA13 FFA12 "v_fmac_f64_dpp %[t], %[xi], %[a12] row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
#define FFA14 FFA13 "v_fmac_f64_dpp %[t], %[xi], %[a13] row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
#define FFA15 FFA14 "v_fmac_f64_dpp %[t], %[xi], %[a14] row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
#define FFA16 FFA15 "v_fmac_f64_dpp %[t], %[xi], %[a15] row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
#define FFB1 "v_fmac_f64_dpp %[xn], %[t], %[b0] row_newbcast:%[c0]+0 row_mask:0xf bank_mask:0xf\n\t"
#define FFB2 FFB1 "v_fmac_f64_dpp %[xn], %[t], %[b1] row_newbcast:%[c0]+1 row_mask:0xf bank_mask:0xf\n\t"
#define FFB3 FFB2 "v_fmac_f64_dpp %[xn], %[t], %[b2] row_newbcast:%[c0]+2 row_mask:0xf bank_mask:0xf\n\t"
#define FFB4 FFB3 "v_fmac_f64_dpp %[xn], %[t], %[b3] row_newbcast:%[c0]+3 row_mask:0xf bank_mask:0xf\n\t"
#define FFB5 FFB4 "v_fmac_f64_dpp %[xn], %[t], %[b4] row_newbcast:%[c0]+4 row_mask:0xf bank_mask:0xf\n\t"
#define FFB6 FFB5 "v_fmac_f64_dpp %[xn], %[t], %[b5] row_newbcast:%[c0]+5 row_mask:0xf bank_mask:0xf\n\t"
#define FFB7 FFB6 "v_fmac_f64_dpp %[xn], %[t], %[b6] row_newbcast:%[c0]+6 row_mask:0xf bank_mask:0xf\n\t"
#define FFB8 FFB7 "v_fmac_f64_dpp %[xn], %[t], %[b7] row_newbcast:%[c0]+7 row_mask:0xf bank_mask:0xf\n\t"
#define FFB9 FFB8 "v_fmac_f64_dpp %[xn], %[t], %[b8] row_newbcast:%[c0]+8 row_mask:0xf bank_mask:0xf\n\t"
#define FFB10 FFB9 "v_fmac_f64_dpp %[xn], %[t], %[b9] row_newbcast:%[c0]+9 row_mask:0xf bank_mask:0xf\n\t"
#define FFB11 FFB10 "v_fmac_f64_dpp %[xn], %[t], %[b10] row_newbcast:%[c0]+10 row_mask:0xf bank_mask:0xf\n\t"
#define FFB12 FFB11 "v_fmac_f64_dpp %[xn], %[t], %[b11] row_newbcast:%[c0]+11 row_mask:0xf bank_mask:0xf\n\t"
#define FFB13 FFB12 "v_fmac_f64_dpp %[xn], %[t], %[b12] row_newbcast:%[c0]+12 row_mask:0xf bank_mask:0xf\n\t"
#define FFB14 FFB13 "v_fmac_f64_dpp %[xn], %[t], %[b13] row_newbcast:%[c0]+13 row_mask:0xf bank_mask:0xf\n\t"
#define FFB15 FFB14 "v_fmac_f64_dpp %[xn], %[t], %[b14] row_newbcast:%[c0]+14 row_mask:0xf bank_mask:0xf\n\t"
#define FFB16 FFB15 "v_fmac_f64_dpp %[xn], %[t], %[b15] row_newbcast:%[c0]+15 row_mask:0xf bank_mask:0xf\n\t"
#define FMA1 [a0] "v"(ma[0])
#define FMA2 FMA1, [a1] "v"(ma[1])
#define FMA3 FMA2, [a2] "v"(ma[2])
#define FMA4 FMA3, [a3] "v"(ma[3])
#define FMA5 FMA4, [a4] "v"(ma[4])
#define FMA6 FMA5, [a5] "v"(ma[5])
#define FMA7 FMA6, [a6] "v"(ma[6])
#define FMA8 FMA7, [a7] "v"(ma[7])
#define FMA9 FMA8, [a8] "v"(ma[8])
#define FMA10 FMA9, [a9] "v"(ma[9])
#define FMA11 FMA10, [a10] "v"(ma[10])
#define FMA12 FMA11, [a11] "v"(ma[11])
#define FMA13 FMA12, [a12] "v"(ma[12])
#define FMA14 FMA13, [a13] "v"(ma[13])
#define FMA15 FMA14, [a14] "v"(ma[14])
#define FMA16 FMA15, [a15] "v"(ma[15])
#define FMB1 [b0] "v"(mb_[0])
#define FMB2 FMB1, [b1] "v"(mb_[1])
#define FMB3 FMB2, [b2] "v"(mb_[2])
#define FMB4 FMB3, [b3] "v"(mb_[3])
#define FMB5 FMB4, [b4] "v"(mb_[4])
#define FMB6 FMB5, [b5] "v"(mb_[5])
#define FMB7 FMB6, [b6] "v"(mb_[6])
#define FMB8 FMB7, [b7] "v"(mb_[7])
#define FMB9 FMB8, [b8] "v"(mb_[8])
#define FMB10 FMB9, [b9] "v"(mb_[9])
#define FMB11 FMB10, [b10] "v"(mb_[10])
#define FMB12 FMB11, [b11] "v"(mb_[11])
#define FMB13 FMB12, [b12] "v"(mb_[12])
#define FMB14 FMB13, [b13] "v"(mb_[13])
#define FMB15 FMB14, [b14] "v"(mb_[14])
#define FMB16 FMB15, [b15] "v"(mb_[15])
#define FUSED_BWD_CASE(NA_, NB_) FUSED_BWD_CASE_(NA_, NB_)
#define FUSED_BWD_CASE_(NA_, NB_)                                                                                       \
    if constexpr (NA == NA_ && NB == NB_) {                                                                             \
        asm("v_add_f64 %[tmp], %[vn], -%[g]\n\t"                                                                        \
            "v_fma_f64 %[qlo], -%[rho], %[tmp], %[qx]\n\t"                                                              \
            "v_fma_f64 %[acc], %[qlo], %[smask], %[cb]\n\t" FCA##NA_ FCB##NB_                                           \
            : [qlo] "=&v"(qlo), [acc] "=&v"(acc), [tmp] "=&v"(tmp)                                                      \
            : [vn] "v"(vn), [g] "v"(g), [qx] "v"(qx), [rho] "v"(rho), [smask] "v"(smask), [cb] "v"(cb), [sa] "v"(sa),    \
              [sb] "v"(sb), [c0] "i"(NA_), FMA##NA_, FMB##NB_);                                                         \
    }
// the same with the cone slack's linear-cost term (admm.cpp:269 | :282 | :295): qlo = fma(-rho, w, fma(-rho, vn - g, qx)), w = vcnew - gc
// (formed once, by the cone step: the W plane of the cone slack cells)
#define FUSED_BWD_SOC_CASE(NA_, NB_) FUSED_BWD_SOC_CASE_(NA_, NB_)
#define FUSED_BWD_SOC_CASE_(NA_, NB_)                                                                                   \
    if constexpr (NA == NA_ && NB == NB_) {                                                                             \
        asm("v_add_f64 %[tmp], %[vn], -%[g]\n\t"                                                                        \
            "v_fma_f64 %[qlo], -%[rho], %[tmp], %[qx]\n\t"                                                              \
            "v_fma_f64 %[qlo], -%[rho], %[w], %[qlo]\n\t"                                                               \
            "v_fma_f64 %[acc], %[qlo], %[smask], %[cb]\n\t" FCA##NA_ FCB##NB_                                           \
            : [qlo] "=&v"(qlo), [acc] "=&v"(acc), [tmp] "=&v"(tmp)                                                      \
            : [vn] "v"(vn), [g] "v"(g), [qx] "v"(qx), [rho] "v"(rho), [smask] "v"(smask), [cb] "v"(cb), [sa] "v"(sa),    \
              [sb] "v"(sb), [w] "v"(w), [c0] "i"(NA_), FMA##NA_, FMB##NB_);                                             \
    }
#define FUSED_FWD_CASE(NA_, NB_) FUSED_FWD_CASE_(NA_, NB_)
#define FUSED_FWD_CASE_(NA_, NB_)                                                                                       \
    if constexpr (NA == NA_ && NB == NB_) {                                                                             \
        asm("v_add_f64 %[tt], %[xi], %[g]\n\t"                                                                          \
            "v_max_f64 %[vm], %[lo], %[tt]\n\t" FFA##NA_                                                                \
            "v_min_f64 %[vn], %[hi], %[vm]\n\t"                                                                         \
            "v_mov_b64 %[xn], %[t]\n\t" FFB##NB_                                                                        \
            : [tt] "=&v"(tt), [vm] "=&v"(vm), [vn] "=&v"(vn), [xn] "=&v"(xn), [t] "+&v"(t)                               \
            : [xi] "v"(xi), [g] "v"(g), [lo] "v"(lo), [hi] "v"(hi), [c0] "i"(NA_), FMA##NA_, FMB##NB_);                  \
    }
// ---- the same fused steps for HALF rows (nx+nu <= 8: two instances per DPP row, see ring1_half): every column is a pair of
// bank-masked FMAs -- all low halves, ONE wait state (a bank-masked DPP op re-writes its disabled lanes with the vdst it read
// without interlock), all high halves.  In the forward step the wait state between the two halves of the first chain is the
// step's own v_min.
#define HAL1 "v_fmac_f64_dpp %[acc], %[sa], %[a0] row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
#define HAL2 HAL1 "v_fmac_f64_dpp %[acc], %[sa], %[a1] row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
#define HAL3 HAL2 "v_fmac_f64_dpp %[acc], %[sa], %[a2] row_newbcast:2 row_mask:0xf bank_mask:0x3\n\t"
#define HAL4 HAL3 "v_fmac_f64_dpp %[acc], %[sa], %[a3] row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
#define HAL5 HAL4 "v_fmac_f64_dpp %[acc], %[sa], %[a4] row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
#define HAL6 HAL5 "v_fmac_f64_dpp %[acc], %[sa], %[a5] row_newbcast:5 row_mask:0xf bank_mask:0x3\n\t"
#define HAL7 HAL6 "v_fmac_f64_dpp %[acc], %[sa], %[a6] row_newbcast:6 row_mask:0xf bank_mask:0x3\n\t"
#define HAL8 HAL7 "v_fmac_f64_dpp %[acc], %[sa], %[a7] row_newbcast:7 row_mask:0xf bank_mask:0x3\n\t"
#define HAH1 "v_fmac_f64_dpp %[acc], %[sa], %[a0] row_newbcast:8+0 row_mask:0xf bank_mask:0xc\n\t"
#define HAH2 HAH1 "v_fmac_f64_dpp %[acc], %[sa], %[a1] row_newbcast:8+1 row_mask:0xf bank_mask:0xc\n\t"
#define HAH3 HAH2 "v_fmac_f64_dpp %[acc], %[sa], %[a2] row_newbcast:8+2 row_mask:0xf bank_mask:0xc\n\t"
#define HAH4 HAH3 "v_fmac_f64_dpp %[acc], %[sa], %[a3] row_newbcast:8+3 row_mask:0xf bank_mask:0xc\n\t"
#define HAH5 HAH4 "v_fmac_f64_dpp %[acc], %[sa], %[a4] row_newbcast:8+4 row_mask:0xf bank_mask:0xc\n\t"
#define HAH6 HAH5 "v_fmac_f64_dpp %[acc], %[sa], %[a5] row_newbcast:8+5 row_mask:0xf bank_mask:0xc\n\t"
#define HAH7 HAH6 "v_fmac_f64_dpp %[acc], %[sa], %[a6] row_newbcast:8+6 row_mask:0xf bank_mask:0xc\n\t"
#define HAH8 HAH7 "v_fmac_f64_dpp %[acc], %[sa], %[a7] row_newbcast:8+7 row_mask:0xf bank_mask:0xc\n\t"
#define HBL1 "v_fmac_f64_dpp %[acc], %[sb], %[b0] row_newbcast:%[c0]+0 row_mask:0xf bank_mask:0x3\n\t"
#define HBL2 HBL1 "v_fmac_f64_dpp %[acc], %[sb], %[b1] row_newbcast:%[c0]+1 row_mask:0xf bank_mask:0x3\n\t"
#define HBL3 HBL2 "v_fmac_f64_dpp %[acc], %[sb], %[b2] row_newbcast:%[c0]+2 row_mask:0xf bank_mask:0x3\n\t"
#define HBL4 HBL3 "v_fmac_f64_dpp %[acc], %[sb], %[b3] row_newbcast:%[c0]+3 row_mask:0xf bank_mask:0x3\n\t"
#define HBL5 HBL4 "v_fmac_f64_dpp %[acc], %[sb], %[b4] row_newbcast:%[c0]+4 row_mask:0xf bank_mask:0x3\n\t"
#define HBL6 HBL5 "v_fmac_f64_dpp %[acc], %[sb], %[b5] row_newbcast:%[c0]+5 row_mask:0xf bank_mask:0x3\n\t"
#define HBL7 HBL6 "v_fmac_f64_dpp %[acc], %[sb], %[b6] row_newbcast:%[c0]+6 row_mask:0xf bank_mask:0x3\n\t"
#define HBL8 HBL7 "v_fmac_f64_dpp %[acc], %[sb], %[b7] row_newbcast:%[c0]+7 row_mask:0xf bank_mask:0x3\n\t"
#define HBH1 "v_fmac_f64_dpp %[acc], %[sb], %[b0] row_newbcast:8+%[c0]+0 row_mask:0xf bank_mask:0xc\n\t"
#define HBH2 HBH1 "v_fmac_f64_dpp %[acc], %[sb], %[b1] row_newbcast:8+%[c0]+1 row_mask:0xf bank_mask:0xc\n\t"
#define HBH3 HBH2 "v_fmac_f64_dpp %[acc], %[sb], %[b2] row_newbcast:8+%[c0]+2 row_mask:0xf bank_mask:0xc\n\t"
#define HBH4 HBH3 "v_fmac_f64_dpp %[acc], %[sb], %[b3] row_newbcast:8+%[c0]+3 row_mask:0xf bank_mask:0xc\n\t"
#define HBH5 HBH4 "v_fmac_f64_dpp %[acc], %[sb], %[b4] row_newbcast:8+%[c0]+4 row_mask:0xf bank_mask:0xc\n\t"
#define HBH6 HBH5 "v_fmac_f64_dpp %[acc], %[sb], %[b5] row_newbcast:8+%[c0]+5 row_mask:0xf bank_mask:0xc\n\t"
#define HBH7 HBH6 "v_fmac_f64_dpp %[acc], %[sb], %[b6] row_newbcast:8+%[c0]+6 row_mask:0xf bank_mask:0xc\n\t"
#define HBH8 HBH7 "v_fmac_f64_dpp %[acc], %[sb], %[b7] row_newbcast:8+%[c0]+7 row_mask:0xf bank_mask:0xc\n\t"
#define HFAL1 "v_fmac_f64_dpp %[t], %[xi], %[a0] row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
#define HFAL2 HFAL1 "v_fmac_f64_dpp %[t], %[xi], %[a1] row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
#define HFAL3 HFAL2 "v_fmac_f64_dpp %[t], %[xi], %[a2] row_newbcast:2 row_mask:0xf bank_mask:0x3\n\t"
#define HFAL4 HFAL3 "v_fmac_f64_dpp %[t], %[xi], %[a3] row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
#define HFAL5 HFAL4 "v_fmac_f64_dpp %[t], %[xi], %[a4] row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
#define HFAL6 HFAL5 "v_fmac_f64_dpp %[t], %[xi], %[a5] row_newbcast:5 row_mask:0xf bank_mask:0x3\n\t"
#define HFAL7 HFAL6 "v_fmac_f64_dpp %[t], %[xi], %[a6] row_newbcast:6 row_mask:0xf bank_mask:0x3\n\t"
#define HFAL8 HFAL7 "v_fmac_f64_dpp %[t], %[xi], %[a7] row_newbcast:7 row_mask:0xf bank_mask:0x3\n\t"
#define HFAH1 "v_fmac_f64_dpp %[t], %[xi], %[a0] row_newbcast:8+0 row_mask:0xf bank_mask:0xc\n\t"
#define HFAH2 HFAH1 "v_fmac_f64_dpp %[t], %[xi], %[a1] row_newbcast:8+1 row_mask:0xf bank_mask:0xc\n\t"
#define HFAH3 HFAH2 "v_fmac_f64_dpp %[t], %[xi], %[a2] row_newbcast:8+2 row_mask:0xf bank_mask:0xc\n\t"
#define HFAH4 HFAH3 "v_fmac_f64_dpp %[t], %[xi], %[a3] row_newbcast:8+3 row_mask:0xf bank_mask:0xc\n\t"
#define HFAH5 HFAH4 "v_fmac_f64_dpp %[t], %[xi], %[a4] row_newbcast:8+4 row_mask:0xf bank_mask:0xc\n\t"
#define HFAH6 HFAH5 "v_fmac_f64_dpp %[t], %[xi], %[a5] row_newbcast:8+5 row_mask:0xf bank_mask:0xc\n\t"
#define HFAH7 HFAH6 "v_fmac_f64_dpp %[t], %[xi], %[a6] row_newbcast:8+6 row_mask:0xf bank_mask:0xc\n\t"
#define HFAH8 HFAH7 "v_fmac_f64_dpp %[t], %[xi], %[a7] row_newbcast:8+7 row_mask:0xf bank_mask:0xc\n\t"
#define HFBL1 "v_fmac_f64_dpp %[xn], %[t], %[b0] row_newbcast:%[c0]+0 row_mask:0xf bank_mask:0x3\n\t"
#define HFBL2 HFBL1 "v_fmac_f64_dpp %[xn], %[t], %[b1] row_newbcast:%[c0]+1 row_mask:0xf bank_mask:0x3\n\t"
#define HFBL3 HFBL2 "v_fmac_f64_dpp %[xn], %[t], %[b2] row_newbcast:%[c0]+2 row_mask:0xf bank_mask:0x3\n\t"
#define HFBL4 HFBL3 "v_fmac_f64_dpp %[xn], %[t], %[b3] row_newbcast:%[c0]+3 row_mask:0xf bank_mask:0x3\n\t"
#define HFBL5 HFBL4 "v_fmac_f64_dpp %[xn], %[t], %[b4] row_newbcast:%[c0]+4 row_mask:0xf bank_mask:0x3\n\t"
#define HFBL6 HFBL5 "v_fmac_f64_dpp %[xn], %[t], %[b5] row_newbcast:%[c0]+5 row_mask:0xf bank_mask:0x3\n\t"
#define HFBL7 HFBL6 "v_fmac_f64_dpp %[xn], %[t], %[b6] row_newbcast:%[c0]+6 row_mask:0xf bank_mask:0x3\n\t"
#define HFBL8 HFBL7 "v_fmac_f64_dpp %[xn], %[t], %[b7] row_newbcast:%[c0]+7 row_mask:0xf bank_mask:0x3\n\t"
#define HFBH1 "v_fmac_f64_dpp %[xn], %[t], %[b0] row_newbcast:8+%[c0]+0 row_mask:0xf bank_mask:0xc\n\t"
#define HFBH2 HFBH1 "v_fmac_f64_dpp %[xn], %[t], %[b1] row_newbcast:8+%[c0]+1 row_mask:0xf bank_mask:0xc\n\t"
#define HFBH3 HFBH2 "v_fmac_f64_dpp %[xn], %[t], %[b2] row_newbcast:8+%[c0]+2 row_mask:0xf bank_mask:0xc\n\t"
#define HFBH4 HFBH3 "v_fmac_f64_dpp %[xn], %[t], %[b3] row_newbcast:8+%[c0]+3 row_mask:0xf bank_mask:0xc\n\t"
#define HFBH5 HFBH4 "v_fmac_f64_dpp %[xn], %[t], %[b4] row_newbcast:8+%[c0]+4 row_mask:0xf bank_mask:0xc\n\t"
#define HFBH6 HFBH5 "v_fmac_f64_dpp %[xn], %[t], %[b5] row_newbcast:8+%[c0]+5 row_mask:0xf bank_mask:0xc\n\t"
#define HFBH7 HFBH6 "v_fmac_f64_dpp %[xn], %[t], %[b6] row_newbcast:8+%[c0]+6 row_mask:0xf bank_mask:0xc\n\t"
#define HFBH8 HFBH7 "v_fmac_f64_dpp %[xn], %[t], %[b7] row_newbcast:8+%[c0]+7 row_mask:0xf bank_mask:0xc\n\t"
// Every wait state such a chain needs is a lane-local instruction the iteration needs anyway (no s_nop): behind the write of the
// accumulator stands the forward constant of the step BEFORE (Dn = fma(p | d, nim, cf) of the source the chain is about to broadcast),
// between the two halves the `vn - g` of the NEXT step; in the forward step `g <- (x + g) - vnew` (the residual terms are NOT pulled in:
// they are only formed when the termination test can pass).  Same instructions, same operands as the one-instance-per-row form: bit-identical results.
#define FUSED_HBWD_CASE(NA_, NB_) FUSED_HBWD_CASE_(NA_, NB_)
#define FUSED_HBWD_CASE_(NA_, NB_)                                                                                      \
    if constexpr (NA == NA_ && NB == NB_) {                                                                             \
        asm("v_fma_f64 %[qlo], -%[rho], %[tmp], %[qx]\n\t"                                                              \
            "v_fma_f64 %[acc], %[qlo], %[smask], %[cb]\n\t"                                                             \
            "v_fma_f64 %[dnp], %[sa], %[nim], %[cf]\n\t" HAL##NA_ HBL##NB_                                              \
            "v_add_f64 %[tmpn], %[vnn], -%[gn]\n\t" HAH##NA_ HBH##NB_                                                   \
            : [qlo] "=&v"(qlo), [acc] "=&v"(acc), [dnp] "=&v"(dnp), [tmpn] "=&v"(tmpn)                                  \
            : [tmp] "v"(tmp), [vnn] "v"(vnn), [gn] "v"(gn), [qx] "v"(qx), [rho] "v"(rho), [smask] "v"(smask), [cb] "v"(cb), \
              [nim] "v"(nim), [cf] "v"(cf), [sa] "v"(sa), [sb] "v"(sb), [c0] "i"(NA_), FMA##NA_, FMB##NB_);             \
    }
#define FUSED_HFWD_CASE(NA_, NB_) FUSED_HFWD_CASE_(NA_, NB_)
#define FUSED_HFWD_CASE_(NA_, NB_)                                                                                      \
    if constexpr (NA == NA_ && NB == NB_) {                                                                             \
        asm("v_add_f64 %[tt], %[xi], %[g]\n\t"                                                                          \
            "v_max_f64 %[vm], %[lo], %[tt]\n\t" HFAL##NA_                                                               \
            "v_min_f64 %[vn], %[hi], %[vm]\n\t" HFAH##NA_                                                               \
            "v_mov_b64 %[xn], %[t]\n\t"                                                                                 \
            "v_add_f64 %[gnew], %[tt], -%[vn]\n\t" HFBL##NB_                                                            \
            "s_nop 0\n\t" HFBH##NB_                                                                                     \
            : [tt] "=&v"(tt), [vm] "=&v"(vm), [vn] "=&v"(vn), [xn] "=&v"(xn), [t] "+&v"(t), [gnew] "=&v"(gnew)           \
            : [xi] "v"(xi), [g] "v"(g), [lo] "v"(lo), [hi] "v"(hi), [c0] "i"(NA_), FMA##NA_, FMB##NB_);                  \
    }
// One (nx, nu) pair per translation unit: the Makefile (compiled-in shapes) and jit.hip (run-time instantiated ones) define
// TINYMPC_FUSED_NX / _NU as plain numbers in front of this header, so that exactly one pair of asm statements is spelled out.
#if defined(TINYMPC_FUSED_NX) && defined(TINYMPC_FUSED_NU)
#define FUSED_SHAPES(CASE) CASE(TINYMPC_FUSED_NX, TINYMPC_FUSED_NU)
constexpr bool fused_shape(int na, int nb) { return na == TINYMPC_FUSED_NX && nb == TINYMPC_FUSED_NU; }
#if TINYMPC_FUSED_NX + TINYMPC_FUSED_NU <= 8
#define FUSED_HALF_SHAPES(CASE) CASE(TINYMPC_FUSED_NX, TINYMPC_FUSED_NU)
#else
#define FUSED_HALF_SHAPES(CASE)
#endif
#else
#define FUSED_SHAPES(CASE)
#define FUSED_HALF_SHAPES(CASE)
constexpr bool fused_shape(int, int) { return false; }
#endif
// backward step: qlo = fma(-rho, vn - g, qx); acc = fma(qlo, smask, cb) + sum_k bcast(sa, k) ma[k] + sum_k bcast(sb, NA + k) mb_[k]
template <int NA, int NB>
__device__ __forceinline__ void fused_backward_step(double& qlo, double& acc, double vn, double g, double qx, double rho, double smask, double cb,
                                                    double sa, double sb, const double* ma, const double* mb_) {
    double tmp;
    FUSED_SHAPES(FUSED_BWD_CASE)
    (void)tmp;
}
template <int NA, int NB>
__device__ __forceinline__ void fused_backward_step_soc(double& qlo, double& acc, double vn, double g, double qx, double w, double rho,
                                                        double smask, double cb, double sa, double sb, const double* ma, const double* mb_) {
    double tmp;
    FUSED_SHAPES(FUSED_BWD_SOC_CASE)
    (void)tmp;
}
// the half-row forms of the two (NA + NB <= 8).  backward: qlo = fma(-rho, tmp, qx) with tmp = vn - g handed in (formed by the step before);
// dnp = fma(sa, nim, cf); tmpn = vnn - gn for the next step.  forward: additionally gnew = tt - vn
template <int NA, int NB>
__device__ __forceinline__ void fused_backward_step_half(double& qlo, double& acc, double& dnp, double& tmpn, double tmp, double vnn, double gn, double qx,
                                                         double rho, double smask, double cb, double nim, double cf, double sa, double sb,
                                                         const double* ma, const double* mb_) {
    FUSED_HALF_SHAPES(FUSED_HBWD_CASE)
}
template <int NA, int NB>
__device__ __forceinline__ void fused_forward_step_half(double& tt, double& vn, double& t, double& xn, double& gnew, double xi, double g,
                                                        double lo, double hi, const double* ma, const double* mb_) {
    double vm;
    FUSED_HALF_SHAPES(FUSED_HFWD_CASE)
    (void)vm;
}
// forward step: tt = xi + g; vn = min(hi, max(lo, tt)); t += sum_k bcast(xi, k) ma[k]; xn = t + sum_k bcast(t, NA + k) mb_[k]
template <int NA, int NB>
__device__ __forceinline__ void fused_forward_step(double& tt, double& vn, double& t, double& xn, double xi, double g, double lo, double hi,
                                                   const double* ma, const double* mb_) {
    double vm;
    FUSED_SHAPES(FUSED_FWD_CASE)
    (void)vm;
}

// v_max_f64 / v_min_f64 without the canonicalising `v_max x, x` hipcc puts in front of fmax/fmin
// operands that come from memory (the box bounds): the hardware quiets NaNs by itself.
__device__ __forceinline__ double vmax64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmin64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max(a, |b|) as ONE opaque instruction (long horizons only, N > 12): spelled with fmax / fabs the N residual updates of a sweep
// form a reduction that the compiler re-associates into a tree BEHIND the sweep, so every slot's x, vnew and x + g stay live to the
// end of the sweep -- at N = 30 that is accumulation-register traffic on every access.  v_max_f64 returns the other operand when
// one is a NaN, as fmax does.
__device__ __forceinline__ double vmax_abs64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool OPAQUE>
__device__ __forceinline__ double resid_max(double a, double b) {
    if constexpr (OPAQUE) return vmax_abs64(a, b);
    else return fmax(a, fabs(b));
}

// The x|u trajectory is written once per launch and never read back by a kernel: it need not displace the warm-start records
// (vnew|znew, g|y, v|z: 240 MiB at 65 536 quadrotor instances) from the 256 MiB Infinity Cache in front of HBM.
// TINYMPC_PRIM_STORE: 0 plain, 1 nontemporal, 2 sc1, 3 sc0 sc1 (experiment builds, tools/build_variants.py)
#ifndef TINYMPC_PRIM_STORE
#define TINYMPC_PRIM_STORE 1
#endif
#ifndef TINYMPC_REF_LOAD
#define TINYMPC_REF_LOAD 1
#endif
// record stores as whole knot segments (see the write-back of admm_solve_kernel): 0 never, 1 the PREFETCH form, 2 every box variant
#ifndef TINYMPC_FULL_LINE_STORES
#define TINYMPC_FULL_LINE_STORES 2
#endif
// per-instance Xref|Uref records are read once per launch and never written by a kernel: the same argument (1 = nontemporal load)
__device__ __forceinline__ double load_ref(const double* p) {
#if TINYMPC_REF_LOAD == 1
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void store_primal(double* p, double v) {
#if TINYMPC_PRIM_STORE == 1
    __builtin_nontemporal_store(v, p);
#elif TINYMPC_PRIM_STORE == 2
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#elif TINYMPC_PRIM_STORE == 3
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
#elif TINYMPC_PRIM_STORE == 4
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}

__device__ __forceinline__ double grp_max16(double v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 16));
    return v;
}
template <int W>
__device__ __forceinline__ double grp_maxw(double v) {                 // max over the W (8 | 16) lanes of an instance
#pragma unroll
    for (int off = W / 2; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, W));
    return v;
}

// project_soc (admm.cpp:39-60) for one component of a 3-cone; s0,s1,s2 = the cone's vector, mine = s[c], the
// component this lane owns.  mu and the norm are float, a/mu is a float division.  The reference is built without
// FMA contraction, and `a` is truncated to float before the branch tests, so the sum of squares must not be fused.
// The two divisions of the "outside" branch are skipped when no lane of the wave is outside its cone, the square root
// when every cone of the wave is clearly inside.
__device__ __forceinline__ double soc_component(double s0, double s1, double s2, double mine, int c, float mu) {
#pragma clang fp contract(off)
    const double u0 = s2 * (double)mu;                                  // :40
    const double q0 = s0 * s0, q1 = s1 * s1;
    const double q = q0 + q1;
    // Fast path: when every cone of the wave is inside with a margin that the float rounding of the norm cannot bridge
    // (sqrt(q) <= u0 (1 - 2^-21) => (float)sqrt(q) <= u0; u0 above 1e-30 and q below 1e70 keep the float normal and finite), the projection is the
    // identity (:49) and the square root is never needed.  NaNs fail the test and take the exact path.
    const bool sure_inside = (u0 > 1e-30) && (q <= (u0 * u0) * (1.0 - 0x1p-20)) && (q < 1e70);
    if (__builtin_amdgcn_ballot_w64(!sure_inside) == 0ull) return mine;
    const float a = (float)sqrt(q);                                     // :42
    const double ad = (double)a;
    const bool below = ad <= -u0, inside = ad <= u0;                    // :46 | :49
    const bool outside = !below && !inside && (ad >= fabs(u0));         // :52 (else :58 -> 0)
    double r = (below || !inside) ? 0.0 : mine;
    if (__builtin_amdgcn_ballot_w64(outside) != 0ull) {
        const double scale = 0.5 * (1.0 + u0 / ad);                     // :55
        const double last = (double)(a / mu);                           // :54
        const double o = scale * ((c == 2) ? last : mine);
        r = outside ? o : r;
    }
    return r;
}

// The same projection for a whole 3-cone held by ONE lane (s0, s1, s2) -> (r0, r1, r2): the transposed form of the cone step
// (one lane per (cone, knot) pair) pays the square root and the two divisions once per iteration instead of once per knot.
// soc_all_inside: EVERY cone of the wave is inside with a margin that the float rounding of the norm cannot bridge (see soc_component):
// every projection of the pass is then the identity (:49) -- the common case of a constraint that is not active -- and the cone step
// neither takes a square root nor rewrites what the forward sweep left.  `&` on purpose: three compares and two s_and, no EXEC regions.
__device__ __forceinline__ bool soc_all_inside(double s0, double s1, double s2, float mu) {
#pragma clang fp contract(off)
    const double u0 = s2 * (double)mu;                                  // :40
    const double q0 = s0 * s0, q1 = s1 * s1;
    const double q = q0 + q1;
    // (u0 < 1e300: a non-finite last component takes the exact path, where gc = (x + gc) - vcnew becomes the NaN the reference gets)
    const bool sure_inside = (u0 > 1e-30) & (u0 < 1e300) & (q <= (u0 * u0) * (1.0 - 0x1p-20)) & (q < 1e70);
    return __builtin_amdgcn_ballot_w64(!sure_inside) == 0ull;
}
// rmu_exact: every mu of the wave is a power of two and rmu = 1 / mu exactly (wave-uniform) -- then a * rmu IS a / mu, rounded once
// like the division (2^k scalings are exact, overflow and gradual underflow included), without the ten instructions of one.
__device__ __forceinline__ void soc_project3(double s0, double s1, double s2, float mu, float rmu, bool rmu_exact, double& r0, double& r1, double& r2) {
#pragma clang fp contract(off)
    const double u0 = s2 * (double)mu;                                  // :40
    const double q0 = s0 * s0, q1 = s1 * s1;
    const double q = q0 + q1;
    const float a = (float)sqrt(q);                                     // :42
    const double ad = (double)a;
    const bool below = ad <= -u0, inside = ad <= u0;                    // :46 | :49
    const bool outside = !below && !inside && (ad >= fabs(u0));         // :52 (else :58 -> 0)
    const bool zero = below || !inside;
    r0 = zero ? 0.0 : s0; r1 = zero ? 0.0 : s1; r2 = zero ? 0.0 : s2;
    if (__builtin_amdgcn_ballot_w64(outside) != 0ull) {
        const double scale = 0.5 * (1.0 + u0 / ad);                     // :55
        const double last = (double)(rmu_exact ? a * rmu : a / mu);      // :54
        r0 = outside ? scale * s0 : r0;
        r1 = outside ? scale * s1 : r1;
        r2 = outside ? scale * last : r2;
    }
}
// (cone, knot) items of a row whose cones start at the lanes of `heads`
__device__ __forceinline__ int soc_items_of(unsigned heads, int nx, int n) {
    int c = 0;
    for (unsigned m = heads; m; m &= m - 1) c += (__builtin_ctz(m) < nx) ? n : n - 1;
    return c;
}
// passes of 16 (cone, knot) items a row needs when every lane triple of the state / input rows carries a cone
constexpr int soc_item_passes(int nx, int nu, int n) { return ((nx / 3) * n + (nu / 3) * (n - 1) + 15) / 16; }

// a + d * b with the product rounded before the sum, as the reference's x86-64 build (no FMA contraction) evaluates
// `Kinf + delta_rho * dKinf_drho` (rho_benchmark.cpp:201-204)
// C1 and C2 are state the iteration never reads (backward_pass_grad works with Quu_inv / AmBKt, admm.cpp:63-77): their Taylor
// steps are logged and applied ADAPT_LOG at a time, in the reference's order (same roundings), instead of one read-modify-write
// of 2 x (nx^2 + nu^2) doubles of HBM per adaptation.
constexpr int ADAPT_LOG = 32;
__device__ __forceinline__ double taylor_step(double a, double d, double b) {
#pragma clang fp contract(off)
    const double t = d * b;
    return a + t;
}

// ---- the kernel -------------------------------------------------------------------------------
// Slot convention: lane j keeps N-long register arrays indexed by "slot" s.  State lanes (j < NX):
// slot s = knot s.  Input lanes: slot s = knot s-1 (slot 0 is a neutral dummy), because the forward
// step i produces x_{i+1} on the state lanes and u_i on the input lanes in the SAME instruction
// stream -- storing both at slot i+1 needs no per-lane select.
// Register budget: 6 N-long FP64 arrays per lane (+2 with a cone) + the matrix rows.  N <= 12 fits the
// 256-VGPR budget of two waves per SIMD; longer horizons take the whole 512-entry file (one wave per SIMD).
// Half-space variants (LIN) take one wave; the adaptive-rho variant keeps two up to N = 10 (its adaptation block, every 5th
// iteration, works out of LDS and spills only on the rare flush_c path).
constexpr int solve_kernel_waves_per_simd(int nz, int n, bool soc, bool lin = false, bool adapt = false) {
    if (lin) return 1;
    if (adapt) return n <= 10 ? 2 : 1;
    return (n <= 10 || 2 * ((soc ? 8 : 6) * n + 2 * nz + 8) + 40 <= 256) ? 2 : 1;
}

// Half-space variants: the slacks live in LDS planes (see the kernel), so a lane holds the box kernel's arrays and the variant takes
// the box kernel's two waves per SIMD -- when eight waves' planes and tables fit the CU's LDS
#ifndef TINYMPC_LIN_WAVES
#define TINYMPC_LIN_WAVES 0                        // (experiments: 1 / 2 instead of the rule)
#endif
// static LDS of a half-space variant with its slack planes: the tables, two planes per set, the cone slack's three, Pinf', the bounds
constexpr long solve_kernel_lin_lds(int nx, int nu, int n, bool soc, int lin, int kmax, bool ub) {
    const int nz = nx + nu, csl = (nz + 1) | 1, cs = (nz + 1) | 1;
    return 8L * (nx * 16 + (ub ? 2 : 2 * n * 16) + ((lin & 1) ? 3 * kmax * 16 + 2 * 4 * n * csl : 0) +
                 ((lin & 2) ? 3 * n * kmax * 16 + 2 * 4 * n * csl : 0) + (soc ? (4 * n + 1) * 3 * cs : 0));
}
// ... which must fit a wave's 64 KiB of static LDS; a variant whose planes do not (long horizons with time-varying tables) keeps its
// slacks in register arrays and projects per knot, as every half-space variant did before round 4 (one wave per SIMD)
constexpr bool solve_kernel_lin_planes(int nx, int nu, int n, bool soc, int lin, int kmax, bool ub) {
    return lin != 0 && solve_kernel_lin_lds(nx, nu, n, soc, lin, kmax, ub) <= 64 * 1024 - 512;
}
constexpr int solve_kernel_lin_waves(int nx, int nu, int n, bool soc, int lin, int kmax, bool ub) {
    if (!solve_kernel_lin_planes(nx, nu, n, soc, lin, kmax, ub)) return 1;
    if (TINYMPC_LIN_WAVES > 0) return TINYMPC_LIN_WAVES;
    return (solve_kernel_waves_per_simd(nx + nu, n, soc) == 2 && 8 * solve_kernel_lin_lds(nx, nu, n, soc, lin, kmax, ub) <= 158 * 1024) ? 2 : 1;
}
// One lane projects whole (knot, family) columns of a half-space slack: column c of instance `inst`, rows [R0, R0 + NF) -- the cells
// pv[(inst * n + c) * csl + R0 ...] hold x + gl; the family's half-spaces are applied one after the other, only when violated
// (admm.cpp:148-173, 186-211; project_hyperplane :70-73; a'z as the reference forms it: products rounded, summed in row order), vlnew
// goes back to pv and gl = (x + gl) - vlnew (:239-254) to pg.  Lane t of the LPI lanes of an instance takes columns S0 + t, S0 + t +
// LPI, ...; tab: [3][KMAX][LW] coefficient | offset | squared norm, + c * tab_stride for the time-varying family.
template <int NF, int KMAX_, int LW_, int LPI_>
__device__ __forceinline__ void project_halfspace_columns(const int t, const int inst, const int n, const int csl, const int R0, const int S0,
                                                          double* pv, double* pg, const double* tab, const int tab_stride, const int nk) {
#pragma clang fp contract(off)
    for (int c = S0 + t; c < n; c += LPI_) {
        const int at = (inst * n + c) * csl + R0;
        const double* tk = tab + c * tab_stride;
        double z[NF], t0[NF];
#pragma unroll
        for (int r = 0; r < NF; ++r) { z[r] = pv[at + r]; t0[r] = z[r]; }
        for (int k = 0; k < nk; ++k) {
            double cv = 0.0;
#pragma unroll
            for (int r = 0; r < NF; ++r) { const double pr = tk[k * LW_ + R0 + r] * z[r]; cv = cv + pr; }
            const double bk = tk[KMAX_ * LW_ + k * LW_ + R0];
            if (cv > bk) {
                const double dist = (cv - bk) / tk[2 * KMAX_ * LW_ + k * LW_ + R0];
#pragma unroll
                for (int r = 0; r < NF; ++r) { const double pr = dist * tk[k * LW_ + R0 + r]; z[r] = z[r] - pr; }
            }
        }
#pragma unroll
        for (int r = 0; r < NF; ++r) { pv[at + r] = z[r]; pg[at + r] = t0[r] - z[r]; }
    }
}

// LIN: bit 0 = static half-spaces (admm.cpp:137-173), bit 1 = time-varying ones (:176-211); 0 = neither
// HET: per-instance problem data (riccati_kernel.hip.h): the matrix rows are re-loaded for every instance
// ADAPT: adaptive rho (admm.cpp:397-423): per-instance rho / Kinf / Pinf, re-estimated every 5th iteration
// UB: the box is the same at every knot (the usual case: constant state / input limits): the two bounds of a lane live in
// registers instead of being read from LDS slot by slot -- 18 LDS reads and as many waits less per iteration at N = 10
// HALF: nx+nu <= 8 -- TWO instances per DPP row (lanes 0-7 | 8-15), eight per wave: every lane-local instruction and every register
// serves twice the instances; a mat-vec column is a pair of bank-masked FMAs (fused_*_step_half).  Plain box variants only.
// PF: the PREFETCH form (SolveArgs::pf_mask): plain box variants, plain launches (no index / perm / trajectory window)
template <int NX, int NU, int N, bool SOC, bool DBG, int MODE, int LIN = 0, bool HET = false, int KMAX = LIN_KMAX, bool ADAPT = false, bool UB = false, bool HALF = false,
          bool PF = false>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(LIN != 0 ? solve_kernel_lin_waves(NX, NU, N, SOC, LIN, KMAX, UB) : solve_kernel_waves_per_simd(NX + NU, N, SOC, false, ADAPT),
                                   LIN != 0 ? solve_kernel_lin_waves(NX, NU, N, SOC, LIN, KMAX, UB) : solve_kernel_waves_per_simd(NX + NU, N, SOC, false, ADAPT))))
void admm_solve_kernel(const SolveArgs P) {
    constexpr bool LS = (LIN & 1) != 0, LT = (LIN & 2) != 0;
    constexpr int NZ = NX + NU;
    static_assert(NZ <= 16, "one instance per 16-lane DPP row");
    constexpr bool FUSED = MODE == 2 && LIN == 0 && fused_shape(NX, NU);              // fused_backward_step(_soc) / fused_forward_step
    // (LIN on the fused blocks -- two / three extra linear-cost terms in the asm statement -- was built and measured: no difference,
    // (12,4,10) + 2 + 2 half-spaces 5.31 ms either way; what the variant pays is its projection step and 208 B/lane of scratch)
    static_assert(!HALF || (NZ <= 8 && FUSED && !SOC && !DBG && !HET && !ADAPT), "half rows: nx+nu <= 8, the plain box kernel on its fused step blocks");
    static_assert(!PF || (MODE == 2 && !SOC && !DBG && LIN == 0 && !HET && !ADAPT), "prefetch form: the plain box kernel");
    constexpr int RL = HALF ? 8 : 16;                                  // lanes of one instance
    constexpr int IPW = 64 / RL;                                       // instances per wave
    constexpr int RECD = N * NZ;                                       // doubles of one instance's record
    constexpr int PF_PIECES = (IPW * RECD * 8 + 1023) / 1024;          // PF: 1-KiB LDS-DMA pieces (64 lanes x 16 B) per array and tile
    constexpr int PF_ARR = PF_PIECES * 128;                            // ... doubles of one array in the tile buffer
    constexpr unsigned long long RMASK = HALF ? 0xFFull : 0xFFFFull;
    const int lane = threadIdx.x & 63;
    const int j = lane & (RL - 1);
    const int grp = lane / RL;
    const bool is_state = j < NX;
    const bool is_input = (j >= NX) && (j < NZ);

    // per-wave LDS copies of the tables that are read with a dynamic index or only once per solve
    __shared__ double sPt[NX * 16];
    __shared__ double sLo[UB ? 1 : N * 16];                    // (UB: the two bounds of a lane live in registers, read from the table once)
    __shared__ double sHi[UB ? 1 : N * 16];
    __shared__ double sLin[LS ? 3 * KMAX * 16 : 1];
    __shared__ double sTLin[LT ? 3 * N * KMAX * 16 : 1];
    // SOC: the cone slack lives in LDS, not in registers.  Per row and slot three planes of CS cells (lane j: cell j; the lanes beyond
    // NZ share cell NZ):
    //   W   what the next backward sweep adds to the linear cost: vcnew - gc (admm.cpp:269 | :282 | :295).  Between the forward
    //       sweep and the cone step it holds x + gc, the vector the cone step projects (:102-109)
    //   GC  gc | yc (:229 / :234)
    //   VC  vcnew | zcnew of the cells that belong to a (cone, knot) item (the other cells: W holds it, see below)
    // The cone step is transposed -- lane j of pass p owns ONE (cone, knot) item: it gathers the item's three components from W,
    // projects them (one square root / division sequence per pass instead of one per knot) and writes all three planes; the sweeps
    // read W (backward) and GC (forward) one step ahead of their use.  Cells outside every item: the projection is the identity,
    // vcnew = x + gc bit for bit, so gc = 0 and W = x + gc -- what the forward sweep leaves there.  A slot is 3 CS doubles, an ODD
    // number: the item gathers of a pass (stride = one slot) fall into distinct LDS banks.  + a dummy slot whose item (0, 0, 1)
    // the lanes without an item project (onto itself): no EXEC-mask region around the gather / scatter of a pass.
    // SPD: how many sweep steps ahead of its use a cell is read (the ring of SPD + 1 registers per plane)
#ifndef TINYMPC_SOC_PD
#define TINYMPC_SOC_PD 2
#endif
    constexpr int SPD = TINYMPC_SOC_PD, SPR = SPD + 1;
    constexpr int CS = SOC ? ((NZ + 1) | 1) : 1;
    constexpr int SLOT_D = 3 * CS;
    constexpr int PL_GC = CS, PL_VC = 2 * CS;
    __shared__ double sC[SOC ? (4 * N + 1) * SLOT_D : 1];
    __shared__ double sP[ADAPT ? 4 * NX * NX : 1];            // ADAPT: each row's own Pinf, column-major (lane j keeps column j current)
    // ADAPT: the lane tables every adaptation reads (ATAB_AT, ATAB_DK, ATAB_DP), lane-major [table][lane][AKC] so that a lane's
    // coefficients are consecutive (ds_read_b128), and each row's log of rho steps that C1 / C2 still have to take (flush_c)
    constexpr int AKC = NX > NU ? NX : NU;
    __shared__ double sTab[ADAPT ? 3 * 16 * AKC : 1];
    __shared__ double sDl[ADAPT ? 4 * ADAPT_LOG : 1];
    if constexpr (ADAPT)
        for (int e = lane; e < 3 * 16 * AKC; e += 64) {
            const int t = e / (16 * AKC), r = e % (16 * AKC);
            sTab[e] = P.atab[t * 256 + (r % AKC) * 16 + r / AKC];              // ATAB_AT / ATAB_DK / ATAB_DP = 0 / 256 / 512
        }
    if constexpr (LS) for (int e = lane; e < 3 * KMAX * 16; e += 64) sLin[e] = P.tab[TAB_BOUNDS + 2 * N * 16 + e];
    if constexpr (LT) for (int e = lane; e < 3 * N * KMAX * 16; e += 64) sTLin[e] = P.tab[TAB_BOUNDS + 2 * N * 16 + 3 * KMAX * 16 + e];
    for (int e = lane; e < NX * 16; e += 64) sPt[e] = P.tab[TAB_PT + e];
    for (int e = lane; e < N * 16; e += 64) {
        if constexpr (!UB) {
            sLo[e] = P.tab[TAB_BOUNDS + e];
            sHi[e] = P.tab[TAB_BOUNDS + N * 16 + e];
        }
    }
    // this lane's matrix rows
    double mb[NZ], mf1[NX], mf2[NU];
#pragma unroll
    for (int k = 0; k < NZ; ++k) mb[k] = P.tab[TAB_MB + k * 16 + j];
#pragma unroll
    for (int k = 0; k < NX; ++k) mf1[k] = P.tab[TAB_MF1 + k * 16 + j];
#pragma unroll
    for (int k = 0; k < NU; ++k) mf2[k] = P.tab[TAB_MF2 + (NX + k) * 16 + j];
    double cb = P.tab[TAB_VEC + VEC_CB * 16 + j];
    double cf = P.tab[TAB_VEC + VEC_CF * 16 + j];
    double qr = P.tab[TAB_VEC + VEC_QR * 16 + j];
    const double smask = P.tab[TAB_VEC + VEC_SMASK * 16 + j];
    const double nim = P.tab[TAB_VEC + VEC_NIM * 16 + j];
    bool soc_lane = false, proj_lane = false;
    int cone_base = -1, cone_c = 0;
    double socmask = 0.0;
    if constexpr (SOC) {
        socmask = P.tab[TAB_VEC + VEC_SOCFLAG * 16 + j];               // 1.0 on the rows that carry a cone slack
        soc_lane = socmask != 0.0;
        cone_base = (int)P.tab[TAB_VEC + VEC_CONE_BASE * 16 + j];
        cone_c = (cone_base >= 0) ? (j - cone_base) : 0;
        proj_lane = soc_lane && cone_base >= 0;
    }
    // Transposed cone step: the (cone, knot) pairs of a row are dealt out to its lanes, 16 per pass -- lane j of pass p takes
    // item 16 p + j, counted cone by cone (ascending base lane): a state cone has N items (slots 0..N-1), an input cone N-1
    // (slots 1..N-1).  All four rows of a wave share the layout.
    constexpr int SOC_PASSES = SOC ? (soc_item_passes(NX, NU, N) > 0 ? soc_item_passes(NX, NU, N) : 1) : 1;
    int item_at[SOC_PASSES];                                   // LDS index (W plane) of the item's first component; the dummy item for lanes without one
    float item_mu[SOC_PASSES], item_rmu[SOC_PASSES];
    bool mu_pow2 = true;                                       // every cone coefficient of the launch is 2^k, |k| <= 60 (wave-uniform): soc_project3 multiplies
    int soc_passes = 0;                                        // passes that hold an item (wave-uniform: a family whose cone is off has none)
    if constexpr (SOC) {
        if (lane < 3) {                                        // the dummy item: W = VC = (0, 0, 1), GC = 0
            sC[4 * N * SLOT_D + lane] = lane == 2 ? 1.0 : 0.0;
            sC[4 * N * SLOT_D + PL_GC + lane] = 0.0;
            sC[4 * N * SLOT_D + PL_VC + lane] = lane == 2 ? 1.0 : 0.0;
        }
        const unsigned heads = (unsigned)(__builtin_amdgcn_ballot_w64(proj_lane && cone_c == 0) & 0xFFFFull);   // row 0 speaks for all
#pragma unroll
        for (int p = 0; p < SOC_PASSES; ++p) {
            int t = p * 16 + j;
            item_at[p] = 4 * N * SLOT_D; item_mu[p] = 1.0f;
            if (heads && p * 16 < soc_items_of(heads, NX, N)) soc_passes = p + 1;
            for (unsigned m = heads; m; m &= m - 1) {
                const int hb = __builtin_ctz(m);
                const int cnt = hb < NX ? N : N - 1;
                if (t >= 0 && t < cnt) {
                    item_at[p] = (grp * N + t + (hb < NX ? 0 : 1)) * SLOT_D + hb;
                    item_mu[p] = (float)P.tab[TAB_VEC + VEC_CONE_MU * 16 + hb];
                    t = -1;
                } else if (t >= 0) t -= cnt;
            }
            item_rmu[p] = 1.0f / item_mu[p];
            const unsigned mb_ = __float_as_uint(item_mu[p]);
            const bool p2 = (mb_ & 0x807FFFFFu) == 0u && (mb_ >> 23) >= 127u - 60u && (mb_ >> 23) <= 127u + 60u;
            if (__builtin_amdgcn_ballot_w64(!p2) != 0ull) mu_pow2 = false;
        }
    }
    bool lin_lane = false, tlin_lane = false;
    if constexpr (LS) lin_lane = P.tab[TAB_VEC + VEC_LINFLAG * 16 + j] != 0.0;
    if constexpr (LT) tlin_lane = P.tab[TAB_VEC + VEC_TLINFLAG * 16 + j] != 0.0;
    // LIN: the half-space slacks live in LDS -- per set (static | time-varying) two planes of one cell per row and slot: V (x + gl
    // between the forward sweep and the projection step, then vlnew) and G (gl); the backward sweep adds -rho (V - G) (admm.cpp:272
    // ...), read two steps ahead like the cone slack.  The projections are transposed (project_halfspace_columns): a lane holds
    // the box kernel's arrays, nothing per slot but one LDS write, and the variant runs two waves per SIMD.
    // (LSP / LTP: the set's slack in planes; LSR / LTR: in register arrays, projected per knot -- the variant whose planes do not fit)
    constexpr bool LPL = solve_kernel_lin_planes(NX, NU, N, SOC, LIN, KMAX, UB);
    constexpr bool LSP = LS && LPL, LTP = LT && LPL, LSR = LS && !LPL, LTR = LT && !LPL;
    constexpr int CSL = LPL ? ((NZ + 1) | 1) : 1;
    __shared__ double sLV[LSP ? 4 * N * CSL : 1], sLG[LSP ? 4 * N * CSL : 1];
    __shared__ double sTV[LTP ? 4 * N * CSL : 1], sTG[LTP ? 4 * N * CSL : 1];
    double ones[(LSR || LTR) ? 16 : 1];
#pragma unroll
    for (int k = 0; k < ((LSR || LTR) ? 16 : 1); ++k) ones[k] = 1.0;
    bool lin_x_on = false, lin_u_on = false, tlin_x_on = false, tlin_u_on = false;     // which families (uniform over the wave)
    if constexpr (LS) { lin_x_on = P.tab[TAB_VEC + VEC_LINFLAG * 16] != 0.0; lin_u_on = P.tab[TAB_VEC + VEC_LINFLAG * 16 + NX] != 0.0; }
    if constexpr (LT) { tlin_x_on = P.tab[TAB_VEC + VEC_TLINFLAG * 16] != 0.0; tlin_u_on = P.tab[TAB_VEC + VEC_TLINFLAG * 16 + NX] != 0.0; }
    double rho = P.rho;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // LDS tables were written by other lanes of this wave

    // UB: slot 1 speaks for every slot (slot 0 of an input lane is the neutral dummy: its box stays (-inf, +inf))
    const double lo_u = P.tab[TAB_BOUNDS + (N > 1 ? 16 : 0) + j], hi_u = P.tab[TAB_BOUNDS + N * 16 + (N > 1 ? 16 : 0) + j];
    const double lo_u0 = P.tab[TAB_BOUNDS + j], hi_u0 = P.tab[TAB_BOUNDS + N * 16 + j];
    // (PF: a plain single-step launch over the whole batch -- no index list, no permutation, no trajectory window: those paths are
    // compiled OUT of the form, because a register-returning load on ANY path between a tile's hand-over and its stores makes the
    // compiler wait for the whole VMEM queue at the join, the next tile's pieces included)
    const int ninst = PF ? P.batch : (P.index ? *P.count : (P.perm ? P.perm_count : P.batch));
    const int ntiles = (ninst + IPW - 1) / IPW;
    const bool resumed = !PF && P.index != nullptr;
    const bool has_traj = !PF && P.traj != nullptr;
    // ---- PREFETCH form: the tile buffer (the launch's dynamic LDS: [x0 piece: 128 doubles][the arrays of pf_mask, PF_ARR each]) and who
    // fills it.  A piece is ONE global_load_lds_dwordx4: lane l's 16 bytes land at piece base + 16 l, so the buffer mirrors the records'
    // own layout [instance][knot][row]; the tile's instances are consecutive (a plain launch).  The last pieces of a tile reach up to
    // 1 KiB past it -- past the END of the array for the last tile: the host allocates every record array and x0 with that much slack
    // (batch_api.hip PF_SLACK_BYTES; nobody reads what those bytes bring).
    extern __shared__ __attribute__((aligned(16))) double sPF[];
    typedef const __attribute__((address_space(1))) void* pf_gptr;
    typedef __attribute__((address_space(3))) void* pf_lptr;
    // Buffer slots (compile-time offsets, so that every buffer read is ONE base register + an immediate): [x0 piece][S0][S1][S2][S3].
    // Warm launches: S0 vnew|znew, S1 g|y, S2 v|z, S3 the reference record when every instance has its own (pf_mask bit 3); cold
    // launches read nothing but the reference record: S0.  EVERY record read of a tile goes through the buffer: one register-returning
    // load left in flight across the hand-over would make the compiler's own wait for it -- which cannot tell the pieces issued behind
    // it apart -- a wait for the next tile's records too, and the prefetch would hide nothing (measured: round 6, first form).
    constexpr int PF_S0 = 128, PF_S1 = PF_S0 + PF_ARR, PF_S2 = PF_S1 + PF_ARR, PF_S3 = PF_S2 + PF_ARR;
    auto pf_issue = [&](const int t) {
        const int tl = P.reverse ? ntiles - 1 - t : t;
        __builtin_amdgcn_global_load_lds((pf_gptr)(P.x0 + ((size_t)tl * (IPW * NX) + (size_t)lane * 2)), (pf_lptr)sPF, 16, 0, 0);
        const size_t t0 = (size_t)tl * (IPW * RECD) + (size_t)lane * 2;
        auto arr = [&](const double* p, const int at) {
#pragma unroll
            for (int q = 0; q < PF_PIECES; ++q) __builtin_amdgcn_global_load_lds((pf_gptr)(p + t0 + q * 128), (pf_lptr)(sPF + at + q * 128), 16, 0, 0);
        };
        if (!P.cold) { arr(P.slack, PF_S0); arr(P.dual, PF_S1); arr(P.slack_prev, PF_S2); }
        if (P.pf_mask & 8) arr(P.ref, P.cold ? PF_S0 : PF_S3);
    };
    unsigned pf_ticket = 0u;                                   // (lane 0) the ticket drawn for the tile after the next one
    const int pf_shard = PF ? (int)(blockIdx.x % (unsigned)(P.pf_shards > 0 ? P.pf_shards : 1)) : 0;
    // (the pointer goes through an empty asm: LLVM's atomic optimizer rewrites an add of a uniform value to a UNIFORM address into one
    // wave-wide add whose result it reads back -- readfirstlane, i.e. a full VMEM wait -- on the spot; a ticket is not needed before the
    // next tile's top, and that is where its wait belongs)
    typedef __attribute__((address_space(1))) unsigned* pf_ctr_t;
    unsigned long long pf_ctr_bits = PF ? (unsigned long long)(P.pf_counter + 16 * pf_shard) : 0ull;
    if constexpr (PF) asm volatile("" : "+v"(pf_ctr_bits));
    const pf_ctr_t pf_ctr = (pf_ctr_t)pf_ctr_bits;
    auto pf_draw = [&]() { return lane == 0 ? __hip_atomic_fetch_add(pf_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; };
    const unsigned pf_base = PF ? P.pf_base[pf_shard] : 0u;
    int pf_next = 0, pf_i = 0;                                 // (pf_i: tiles this wave has taken)
    bool pf_first = true;
    // a reference record every instance shares: -(ref x diagonal) and the terminal term are formed ONCE per wave and kept in LDS
    // (one 16-lane row per slot): a tile reads them back instead of carrying ten more doubles per lane from tile to tile
    double* const sQX = sPF + PF_S3;                           // (the slot of the per-instance records: free when the record is shared)
    if constexpr (PF) {
        if ((int)blockIdx.x < ntiles) {
            pf_issue((int)blockIdx.x);
            if (P.pf_static <= 1) pf_ticket = pf_draw();           // (the second tile is a ticketed one already)
        }
        if (P.ref_shared) {
            double rl = 0.0, qxs[N];
#pragma unroll
            for (int s = 0; s < N; ++s) {
                const bool valid = is_state || (is_input && s >= 1);
                const double r = valid ? P.ref[j - (is_input ? NZ : 0) + s * NZ] : 0.0;
                qxs[s] = -(r * qr);
                if (s == N - 1) rl = r;
            }
            double pt[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) pt[k] = sPt[k * 16 + j];
            double xp = 0.0;
            if constexpr (HALF) ring1_half<0, NX>(xp, rl, pt);
            else xp = ring_sum<MODE, 0, NX>(0.0, rl, pt);
            qxs[N - 1] = is_state ? -xp : qxs[N - 1];          // (terminal_term below, for the shared record)
            if (grp == 0) {
#pragma unroll
                for (int s = 0; s < N; ++s) sQX[s * 16 + j] = qxs[s];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // Tiles of 4 instances: one per wave (grid = tiles), or -- a follow-up stage of a split solve, fewer waves than tiles -- the wave
    // takes its next tile off the stage's counter the moment it is free (its tiles differ in depth: a fixed stride would make the
    // stage wait for the slot that drew the deepest ones)
    for (int tile = blockIdx.x; tile < ntiles;
         tile = PF ? pf_next : (P.work_counter ? __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(P.work_counter, 1) : 0) + (int)gridDim.x : tile + (int)gridDim.x)) {
        const int slot = (P.reverse ? ntiles - 1 - tile : tile) * IPW + grp;
        // PF: every lane takes its part of the tile out of the buffer (the rows of a last, partial tile: whatever the clamped pieces
        // brought -- they do not use it), then the buffer is handed to the NEXT tile's records: issued here, under the full EXEC mask,
        // in flight while this tile iterates.  Order of this wave's VMEM operations per tile: [reads of the arrays that do not travel
        // through the buffer] [ticket] [the next tile's pieces] ... [this tile's stores].
        double pfVN[PF ? N : 1], pfG[PF ? N : 1], pfVP[PF ? N : 1], pfQXs[PF ? N : 1], pf_x0 = 0.0, pf_rl = 0.0;
        if constexpr (PF) {
            // The tile's records have landed when the wave's own VMEM counter says so -- nothing else orders a ds_read behind a pending
            // LDS-DMA.  A wave's VMEM operations complete in issue order; behind this tile's pieces only the stores of the tile before
            // were issued, among them N each to the vnew|znew and g|y records when the launch writes both: those may stay in flight.
            if (pf_first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if ((P.store_mask & 6) == 6) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * N <= 63 ? 2 * N : 63) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pf_first = false;
            if (pf_i + 1 < P.pf_static) pf_next = tile + (int)gridDim.x;
            else {
                unsigned pf_t;                                    // (volatile: the read-back stays HERE, behind the wait above)
                asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(pf_t) : "v"(pf_ticket));
                pf_next = (int)(pf_t - pf_base) * P.pf_shards + pf_shard + (P.pf_static > 1 ? P.pf_static : 1) * (int)gridDim.x;
            }
            const int m = P.pf_mask;
            // this lane's cell of slot 0 in a buffer array (input lanes: knot s-1 at slot s; their slot 0 and the lanes beyond nx+nu read
            // a cell nobody uses: at worst the 16 doubles in FRONT of the array, which the x0 piece covers)
            const int lrow = grp * RECD + (j < NZ ? j - (is_input ? NZ : 0) : 0);
            const bool warm_t = !P.cold;
            pf_x0 = sPF[grp * NX + (is_state ? j : 0)];
#pragma unroll
            for (int s = 0; s < N; ++s) { pfVN[s] = 0.0; pfG[s] = 0.0; pfVP[s] = 0.0; }
            if (warm_t) {
#pragma unroll
                for (int s = 0; s < N; ++s) { pfVN[s] = sPF[PF_S0 + lrow + s * NZ]; pfG[s] = sPF[PF_S1 + lrow + s * NZ]; }
            }
            auto own_ref = [&](const int at) {                    // this instance's own reference record: -(ref x diagonal), admm.cpp:266 / :279
#pragma unroll
                for (int s = 0; s < N; ++s) {
                    const bool valid = is_state || (is_input && s >= 1);
                    const double r = valid ? sPF[at + lrow + s * NZ] : 0.0;
                    pfQXs[s] = -(r * qr);
                    if (s == N - 1) pf_rl = r;
                }
            };
            if (m & 8) { if (warm_t) own_ref(PF_S3); else own_ref(PF_S0); }      // (two call sites: the slot is an immediate in each)
            else {
#pragma unroll
                for (int s = 0; s < N; ++s) pfQXs[s] = sQX[s * 16 + j];
            }
            if (warm_t) {
#pragma unroll
                for (int s = 0; s < N; ++s) pfVP[s] = sPF[PF_S2 + lrow + s * NZ];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer's contents are in registers: it is free
            if (pf_next < ntiles) {                               // (a wave stops drawing with its first ticket beyond the batch)
                // the tile after the next one is a ticketed one.  A BRANCH (the empty asm keeps it one): as a select, or assigned on every
                // path, the ticket register -- which may have a draw in flight -- would be touched in every tile, and the compiler waits
                // for the whole VMEM queue before it touches it: the stores of the tile before would never stay in flight
                if (pf_i + 2 >= P.pf_static) {
                    asm volatile("" ::: "memory");
                    pf_ticket = pf_draw();
                }
                pf_issue(pf_next);
            }
            ++pf_i;
        }
        (void)pfVN; (void)pfG; (void)pfVP; (void)pf_rl; (void)pf_x0; (void)pfQXs; (void)pf_first; (void)pf_ticket; (void)sQX; (void)pf_ctr; (void)pf_base; (void)pf_draw; (void)pf_i;
        if (slot < ninst) {
            const int b = PF ? slot : (resumed ? P.index[slot] : (P.perm ? P.perm[slot] : slot));
            const double* het = nullptr;
            if constexpr (HET) {                               // this instance's own cache (A, B, Q, R, rho differ per instance)
                het = P.het_tabs + (size_t)b * TAB_BOUNDS;
#pragma unroll
                for (int k = 0; k < NZ; ++k) mb[k] = het[TAB_MB + k * 16 + j];
#pragma unroll
                for (int k = 0; k < NX; ++k) mf1[k] = het[TAB_MF1 + k * 16 + j];
#pragma unroll
                for (int k = 0; k < NU; ++k) mf2[k] = het[TAB_MF2 + (NX + k) * 16 + j];
                cb = het[TAB_VEC + VEC_CB * 16 + j];
                cf = het[TAB_VEC + VEC_CF * 16 + j];
                qr = het[TAB_VEC + VEC_QR * 16 + j];
                rho = het[TAB_VEC + 8 * 16 + j];               // VEC_RHO (riccati_kernel.hip.h)
            }
            if constexpr (ADAPT) {                             // this instance's own rho / Kinf / Pinf (they persist from solve to solve)
                rho = P.arho[b];
                const double* aK = P.aK + (size_t)b * (NU * NX);
                if (is_state) {
#pragma unroll
                    for (int k = 0; k < NU; ++k) mb[NX + k] = -aK[k + NU * j];            // -Kinf'[j][k]
#pragma unroll
                    for (int k = 0; k < NX; ++k) sP[grp * NX * NX + k + NX * j] = P.aP[(size_t)b * (NX * NX) + k + NX * j];
                } else if (is_input) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) mf1[k] = -aK[(j - NX) + NU * k];         // -Kinf[j-nx][k]
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // record base of this lane: input lanes read knot s-1 at slot s
            const size_t lbase = (size_t)b * (N * NZ) + j - (is_input ? NZ : 0);
            double X[N], G[N], VN[N], VP[N], QX[N], Dn[N - 1];
            // cone slack cells of this lane (W plane of slot 0): sC[cw + s * SLOT_D (+ PL_GC | PL_VC)]
            const int cw = grp * N * SLOT_D + (j < NZ ? j : NZ);
            const double x0v_in = is_state ? (PF ? pf_x0 : P.x0[(size_t)b * NX + j]) : 0.0;      // tiny_set_x0
            const int cl = grp * N * CSL + (j < NZ ? j : NZ);  // LIN: this lane's cell of slot 0 (slot s: + s * CSL)
            double VL[LSR ? N : 1], GL[LSR ? N : 1], VT[LTR ? N : 1], GT[LTR ? N : 1];
            double Qd[DBG ? N : 1], Pd[DBG ? N : 1], Dd[DBG ? N : 1];
            double ref_last = 0.0, qx_last_plain = 0.0;
            // ---- load the instance record (coalesced: contiguous NZ*8-byte knot segments)
#pragma unroll
            for (int s = 0; s < N; ++s) {
                const bool valid = is_state || (is_input && s >= 1);
                const size_t off = lbase + s * NZ;
                const bool warm = valid && !P.cold;
                if constexpr (PF) {                          // (what the tile buffer held; a shared reference record: formed once per wave)
                    VN[s] = warm ? pfVN[s] : 0.0;
                    G[s] = warm ? pfG[s] : 0.0;
                    VP[s] = warm ? pfVP[s] : 0.0;
                    QX[s] = pfQXs[s];
                    X[s] = 0.0;
                    if (s == N - 1) ref_last = pf_rl;
                    continue;
                }
                const double r = valid ? (P.ref_shared ? P.ref[off - (size_t)b * (N * NZ)] : load_ref(P.ref + off)) : 0.0;
                VN[s] = warm ? P.slack[off] : 0.0;
                G[s] = warm ? P.dual[off] : 0.0;
                VP[s] = warm ? P.slack_prev[off] : 0.0;
                QX[s] = -(r * qr);                           // admm.cpp:266 / :279
                X[s] = 0.0;
                if (s == N - 1) ref_last = r;
                if constexpr (SOC) {
                    // vcnew = x, zcnew = u of the solve before (admm.cpp:352-357; x[:,0] = x0 is already in place then); gc | yc
                    double vc0 = (warm && soc_lane) ? (resumed ? P.cslack : P.prim)[off] : 0.0;
                    const double gc0 = (warm && soc_lane) ? P.cdual[off] : 0.0;
                    if (s == 0 && is_state && soc_lane && !resumed) vc0 = x0v_in;
                    sC[cw + s * SLOT_D] = vc0 - gc0;
                    sC[cw + s * SLOT_D + PL_GC] = gc0;
                    sC[cw + s * SLOT_D + PL_VC] = vc0;
                }
                if constexpr (LS) {
                    const double vl0 = (warm && lin_lane) ? (resumed ? P.lslack : P.prim)[off] : 0.0;      // admm.cpp:361-365
                    const double gl0 = (warm && lin_lane) ? P.ldual[off] : 0.0;
                    if constexpr (LSP) { sLV[cl + s * CSL] = vl0; sLG[cl + s * CSL] = gl0; } else { VL[s] = vl0; GL[s] = gl0; }
                }
                if constexpr (LT) {
                    const double vt0 = (warm && tlin_lane) ? (resumed ? P.tlslack : P.prim)[off] : 0.0;    // admm.cpp:370-374
                    const double gt0 = (warm && tlin_lane) ? P.tldual[off] : 0.0;
                    if constexpr (LTP) { sTV[cl + s * CSL] = vt0; sTG[cl + s * CSL] = gt0; } else { VT[s] = vt0; GT[s] = gt0; }
                }
                if constexpr (DBG) { Qd[s] = 0.0; Pd[s] = 0.0; Dd[s] = 0.0; }
            }
            double x0v = x0v_in;
            auto terminal_term = [&]() {   // -(Xref[:,N-1]^T Pinf) (admm.cpp:292); only state lanes' ref_last is broadcast
                double pt[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k)
                    pt[k] = ADAPT ? sP[grp * NX * NX + k + NX * (is_state ? j : 0)] : (HET ? het[TAB_PT + k * 16 + j] : sPt[k * 16 + j]);
                double xp = 0.0;
                if constexpr (HALF) ring1_half<0, NX>(xp, ref_last, pt);
                else xp = ring_sum<MODE, 0, NX>(0.0, ref_last, pt);
                qx_last_plain = QX[N - 1];                   // q[:,N-1] uses -Xref*Q, p[:,N-1] the terminal term
                QX[N - 1] = is_state ? -xp : QX[N - 1];
            };
            int ndl = 0;                                       // ADAPT: rho steps logged in sDl and not yet taken by C1 / C2
            auto flush_c = [&](int n) {                        // update_matrices_with_derivatives, rho_benchmark.cpp:196-210 (C1, C2)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (P.aC2 && is_state) {
                    double* c2 = P.aC2 + (size_t)b * (NX * NX) + NX * j;
                    double v[NX], d[NX];
#pragma unroll
                    for (int k = 0; k < NX; ++k) { v[k] = c2[k]; d[k] = P.atab[ATAB_DC2 + k * 16 + j]; }
                    for (int i = 0; i < n; ++i) {
                        const double dl = sDl[grp * ADAPT_LOG + i];
#pragma unroll
                        for (int k = 0; k < NX; ++k) v[k] = taylor_step(v[k], dl, d[k]);
                    }
#pragma unroll
                    for (int k = 0; k < NX; ++k) c2[k] = v[k];
                }
                if (P.aC1 && j < NU) {
                    double* c1 = P.aC1 + (size_t)b * (NU * NU) + NU * j;
                    double v[NU], d[NU];
#pragma unroll
                    for (int k = 0; k < NU; ++k) { v[k] = c1[k]; d[k] = P.atab[ATAB_DC1 + k * 16 + j]; }
                    for (int i = 0; i < n; ++i) {
                        const double dl = sDl[grp * ADAPT_LOG + i];
#pragma unroll
                        for (int k = 0; k < NU; ++k) v[k] = taylor_step(v[k], dl, d[k]);
                    }
#pragma unroll
                    for (int k = 0; k < NU; ++k) c1[k] = v[k];
                }
            };
            if (!has_traj && !(PF && P.ref_shared)) terminal_term();
            const int traj_k0 = has_traj ? (P.traj_step0 + (P.traj_offsets ? P.traj_offsets[b] : 0)) : 0;

            int iter = 0, solved = 0, checked = 0;
            unsigned acc_iter = 0, acc_solved = 0;
            bool vp_touched = false;                           // v|z differ from what was loaded (a solve that converges at its first check leaves them alone)
            double rp = 0.0, rd = 0.0;
            const int nsteps = PF ? 1 : (P.steps > 1 ? P.steps : 1);
            const int iter_first = resumed ? P.iter_base : 0;  // (a multiple of check_termination: the countdown restarts in phase)
            for (int step = 0; step < nsteps; ++step) {        // closed-loop MPC steps fused in one launch
                X[0] = x0v;                                    // work->x.col(0) = x0
                if (has_traj) {                                // work->Xref = Xref_total.block(0, k, nx, N)
                    if (is_state) {
#pragma unroll
                        for (int s = 0; s < N; ++s) {
                            int kk = traj_k0 + step + s;
                            kk = kk < P.traj_points ? kk : P.traj_points - 1;
                            const double r = P.traj[(size_t)kk * NX + j];
                            QX[s] = -(r * qr);
                            if (s == N - 1) ref_last = r;
                        }
                    }
                    terminal_term();
                }
                if (P.reset_duals) {                           // work->y = 0; work->g = 0
#pragma unroll
                    for (int s = 0; s < N; ++s) G[s] = 0.0;
                }
                if constexpr (SOC) {
                    if (step > 0) {                            // vcnew = x, zcnew = u of the previous solve (admm.cpp:352-357)
#pragma unroll
                        for (int s = 0; s < N; ++s) {
                            const double vc0 = soc_lane ? X[s] : 0.0;
                            sC[cw + s * SLOT_D] = vc0 - sC[cw + s * SLOT_D + PL_GC];
                            sC[cw + s * SLOT_D + PL_VC] = vc0;
                        }
                    }
                }
                if constexpr (LS) {                            // vlnew = x, zlnew = u (admm.cpp:361-365)
                    if (step > 0) {
#pragma unroll
                        for (int s = 0; s < N; ++s) { if constexpr (LSP) sLV[cl + s * CSL] = lin_lane ? X[s] : 0.0; else VL[s] = lin_lane ? X[s] : 0.0; }
                    } else if (is_state && lin_lane && !resumed) { if constexpr (LSP) sLV[cl] = x0v; else VL[0] = x0v; }
                }
                if constexpr (LT) {                            // vlnew_tv = x, zlnew_tv = u (admm.cpp:370-374)
                    if (step > 0) {
#pragma unroll
                        for (int s = 0; s < N; ++s) { if constexpr (LTP) sTV[cl + s * CSL] = tlin_lane ? X[s] : 0.0; else VT[s] = tlin_lane ? X[s] : 0.0; }
                    } else if (is_state && tlin_lane && !resumed) { if constexpr (LTP) sTV[cl] = x0v; else VT[0] = x0v; }
                }
                if constexpr (LPL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const int iter0 = iter_first;
                iter = iter0; solved = 0;
                if (resumed && P.check_termination > 0) checked = 1;
                int countdown = P.check_termination;
                // SOC: vcnew - gc of slot i comes out of the W plane two sweep steps before its use (a ring of three registers); the
                // first two of a sweep are read at the end of the iteration before
                double wr[SOC ? SPR : 1];
                bool gcz[SOC_PASSES];                          // SOC: the GC cells of pass p's items are known to be zero (rows only LEAVE a solve)
#pragma unroll
                for (int p = 0; p < SOC_PASSES; ++p) gcz[p] = false;
                if constexpr (SOC) {
#pragma unroll
                    for (int d = 1; d <= SPR && d <= N; ++d) wr[(N - d) % SPR] = sC[cw + (N - d) * SLOT_D];
                }
                for (int it = iter0; it < P.max_iter; ++it) {
                    // LIN: vlnew and gl of slot i come out of their planes two sweep steps before their use (rings of three registers)
                    double lvr[LSP ? SPR : 1], lgr[LSP ? SPR : 1], tvr[LTP ? SPR : 1], tgr[LTP ? SPR : 1];
                    if constexpr (LPL) {
#pragma unroll
                        for (int d = 1; d <= SPR && d <= N; ++d) {
                            if constexpr (LSP) { lvr[(N - d) % SPR] = sLV[cl + (N - d) * CSL]; lgr[(N - d) % SPR] = sLG[cl + (N - d) * CSL]; }
                            if constexpr (LTP) { tvr[(N - d) % SPR] = sTV[cl + (N - d) * CSL]; tgr[(N - d) % SPR] = sTG[cl + (N - d) * CSL]; }
                        }
                    }
                    // vlnew - gl | vlnew_tv - gl_tv of slot s_: from the rings, or the register arrays of the per-knot variant
                    auto lin_w = [&](const int s_) { if constexpr (LSP) return lvr[s_ % SPR] - lgr[s_ % SPR]; else return VL[LSR ? s_ : 0] - GL[LSR ? s_ : 0]; };
                    auto tlin_w = [&](const int s_) { if constexpr (LTP) return tvr[s_ % SPR] - tgr[s_ % SPR]; else return VT[LTR ? s_ : 0] - GT[LTR ? s_ : 0]; };
                    // ---- update_linear_cost (lane-local) fused into the backward sweep.
                    // qv(s): state lanes q_s (s = N-1: the terminal p), input lanes r_{s-1}.
                    double qhi;
                    {
                        double t = fma(-rho, VN[N - 1] - G[N - 1], QX[N - 1]);      // admm.cpp:293 | :280
                        if constexpr (SOC) t = fma(-rho, wr[(N - 1) % SPR], t);     // :295 | :282
                        if constexpr (LS) t = fma(-rho, lin_w(N - 1), t);           // :298 | :285
                        if constexpr (LT) t = fma(-rho, tlin_w(N - 1), t);          // :301 | :288
                        qhi = t;
                        if constexpr (DBG) {
                            double ql = fma(-rho, VN[N - 1] - G[N - 1], qx_last_plain);   // q[:,N-1], :267
                            if constexpr (SOC) ql = fma(-rho, wr[(N - 1) % SPR], ql);       // :269
                            if constexpr (LS) ql = fma(-rho, lin_w(N - 1), ql);             // :272
                            if constexpr (LT) ql = fma(-rho, tlin_w(N - 1), ql);            // :275
                            Qd[N - 1] = is_state ? ql : t;
                            Pd[N - 1] = t;
                        }
                    }
                    double pcur = qhi;                         // p_{N-1} on state lanes
                    // ---- backward_pass_grad, admm.cpp:13-20
                    if constexpr (HALF) {                       // (the wait states of the half-row chains carry the neighbouring steps' lane-local work)
                        double tmpv = VN[N - 2] - G[N - 2];
#pragma unroll
                        for (int i = N - 2; i >= 0; --i) {
                            double qlo, res, dnp, tmpn;
                            constexpr int dummy = 0;
                            const int in = i > 0 ? i - 1 : dummy;
                            fused_backward_step_half<NX, NU>(qlo, res, dnp, tmpn, tmpv, VN[in], G[in], QX[i], rho, smask, cb, nim, cf, pcur, qhi, mb, mb + NX);
                            if (i + 1 <= N - 2) Dn[i + 1] = dnp;                        // fma(p_{i+1} | d_{i+1}, nim, cf)
                            tmpv = tmpn;
                            pcur = res;                                                 // p_i | d_i
                            qhi = qlo;
                        }
                        Dn[0] = fma(pcur, nim, cf);
                    }
#pragma unroll
                    for (int i = N - 2; i >= 0 && !HALF; --i) {
                        if constexpr (SOC) {
                            if (i >= SPD) wr[(i - SPD) % SPR] = sC[cw + (i - SPD) * SLOT_D];      // (slot i + 1's register is free by now)
                        }
                        if constexpr (LSP) { if (i >= SPD) { lvr[(i - SPD) % SPR] = sLV[cl + (i - SPD) * CSL]; lgr[(i - SPD) % SPR] = sLG[cl + (i - SPD) * CSL]; } }
                        if constexpr (LTP) { if (i >= SPD) { tvr[(i - SPD) % SPR] = sTV[cl + (i - SPD) * CSL]; tgr[(i - SPD) % SPR] = sTG[cl + (i - SPD) * CSL]; } }
                        if constexpr (SOC || LPL) __builtin_amdgcn_sched_barrier(0);
                        if constexpr (FUSED) {                  // linear-cost terms + both mat-vec chains in one asm statement (no s_nop)
                            double qlo, res;
                            if constexpr (SOC) fused_backward_step_soc<NX, NU>(qlo, res, VN[i], G[i], QX[i], wr[i % SPR], rho, smask, cb, pcur, qhi, mb, mb + NX);
                            else fused_backward_step<NX, NU>(qlo, res, VN[i], G[i], QX[i], rho, smask, cb, pcur, qhi, mb, mb + NX);
                            pcur = res;                                                 // p_i | d_i
                            Dn[i] = fma(res, nim, cf);
                            if constexpr (DBG) { Qd[i] = qlo; Pd[i] = res; Dd[i + 1] = res; }
                            qhi = qlo;
                            continue;
                        }
                        double qlo = fma(-rho, VN[i] - G[i], QX[i]);                // :267 | :280
                        if constexpr (SOC) qlo = fma(-rho, wr[i % SPR], qlo);       // :269 | :282
                        if constexpr (LS) qlo = fma(-rho, lin_w(i), qlo);           // :272 | :285
                        if constexpr (LT) qlo = fma(-rho, tlin_w(i), qlo);          // :275 | :288
                        // state lanes: q_i + APf + AmBKt p_{i+1} - Kinf' r_i ; input lanes: Quu_inv (B' p_{i+1} + r_i + BPf)
                        const double res = ring_sum2<MODE, NX, NU>(fma(qlo, smask, cb), pcur, mb, qhi, mb + NX);
                        pcur = res;                                                 // p_i | d_i
                        Dn[i] = fma(res, nim, cf);                                  // input lanes: -d_i; state lanes: fdyn (the forward step's constant)
                        if constexpr (DBG) { Qd[i] = qlo; Pd[i] = res; Dd[i + 1] = res; }
                        qhi = qlo;
                    }
                    // ---- forward_pass (admm.cpp:25-32) with update_slack + update_dual + the residual maxima
                    // (admm.cpp:81-135, 219-235, 314-317) of slot i+1 issued right behind the step that produced it:
                    // the lane-local element-wise work (and its LDS bound reads) fills the dependency stalls of
                    // the next step's FMA chain instead of forming a separate, latency-exposed phase.
                    // The 2N subtractions and maxima behind the residuals: short horizons form them inside the termination test, from the
                    // registers, only when the probe lets the test run (said here, not left to the compiler's sinking); long ones keep
                    // the running maxima of the forward sweep
                    constexpr bool LAZY_RES = N <= 12;
                    double pmax = 0.0, dmax = 0.0;
                    auto slot_update = [&](const int s, const double lo, const double hi, const double gcv, const double glv, const double gtv) {
                        const double xi = X[s];
                        const double t = xi + G[s];                                 // :85 / :88
                        const double vn = vmin64(hi, vmax64(lo, t));                // :91-98
                        if constexpr (!LAZY_RES) {
                            pmax = resid_max<true>(pmax, xi - vn);
                            dmax = resid_max<true>(dmax, VP[s] - vn);
                        }
                        G[s] = t - vn;                      // :222 / :225  g + x - vnew; (g + x) == t bit-for-bit
                        VN[s] = vn;
                        if constexpr (SOC) {
                            // vcnew = x + gc on every row of a family whose cone slack is on (:102-109); gc is 0 on the
                            // other rows, so one FMA against the 0/1 mask does the add and the select.  Projected after the
                            // sweep, one lane per (cone, knot): the cone step below; a cell outside every item keeps this
                            // value as its vcnew, so its gc = (x + gc) - vcnew = 0 (written once per solve, behind the cone step)
                            sC[cw + s * SLOT_D] = fma(xi, socmask, gcv);
                        }
                        // half-space slacks: vlnew = x + gl on the rows of a family whose slack is on (:139 / :144 / :177 / :182), projected
                        // after the sweep, one lane per (knot, family) column (project_halfspace_columns)
                        if constexpr (LSP) sLV[cl + s * CSL] = (lin_lane && (is_state || s >= 1)) ? (xi + glv) : 0.0;
                        if constexpr (LTP) sTV[cl + s * CSL] = (tlin_lane && (is_state || s >= 1)) ? (xi + gtv) : 0.0;
                        if constexpr (LSR || LTR) {
                            // the per-knot form: a'z is a lane-local product summed over the row with the broadcast-FMA chain (against a vector of
                            // ones), separately for the state and the input rows; constraints are applied sequentially, only when violated (:154)
                            auto halfspaces = [&](double z, const double* tabk, const int nk) {
                                for (int k = 0; k < nk; ++k) {
                                    const double a = tabk[k * 16 + j];
                                    const double bk = tabk[KMAX * 16 + k * 16 + j];
                                    const double nn = tabk[2 * KMAX * 16 + k * 16 + j];
                                    const double prod = a * z;
                                    double cs = 0.0, ci = 0.0;
                                    ring1<0, NX>(cs, prod, ones);
                                    ring1<NX, NU>(ci, prod, ones);
                                    const double cv = is_state ? cs : ci;
                                    if (cv > bk) z = z - ((cv - bk) / nn) * a;
                                }
                                return z;
                            };
                            if constexpr (LSR) {
                                const bool on = lin_lane && (is_state || s >= 1);
                                double vl = on ? (xi + GL[s]) : 0.0;                    // :139 / :144
                                vl = halfspaces(vl, sLin, P.n_lin);
                                GL[s] = on ? ((GL[s] + xi) - vl) : 0.0;                 // :239 / :244
                                VL[s] = on ? vl : 0.0;
                            }
                            if constexpr (LTR) {
                                const bool on = tlin_lane && (is_state || s >= 1);
                                double vt = on ? (xi + GT[s]) : 0.0;                    // :177 / :182
                                vt = halfspaces(vt, sTLin + s * 3 * KMAX * 16, P.n_tlin);
                                GT[s] = on ? ((GT[s] + xi) - vt) : 0.0;                 // :249 / :254
                                VT[s] = on ? vt : 0.0;
                            }
                        }
                    };
                    // Software pipeline: the box bounds of slot i+1 are read from LDS one whole step before
                    // they are used (sched_barrier pins the reads above the step), and slot i's update is
                    // scheduled together with the FMA chain of step i -- both only need x_i.
                    double lo_c = UB ? lo_u0 : sLo[j], hi_c = UB ? hi_u0 : sHi[j];
                    double gr[SOC ? SPR : 1];                   // SOC: gc of slot i, read from its plane SPD steps ahead
                    if constexpr (SOC) {
#pragma unroll
                        for (int d = 0; d < SPD && d < N; ++d) gr[d % SPR] = sC[cw + d * SLOT_D + PL_GC];
                    } else gr[0] = 0.0;
                    double glr[LSP ? SPR : 1], gtr[LTP ? SPR : 1];   // LIN: gl | gl_tv of slot i, likewise
                    glr[0] = 0.0; gtr[0] = 0.0;
                    if constexpr (LPL) {
#pragma unroll
                        for (int d = 0; d < SPD && d < N; ++d) {
                            if constexpr (LSP) glr[d % SPR] = sLG[cl + d * CSL];
                            if constexpr (LTP) gtr[d % SPR] = sTG[cl + d * CSL];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < N - 1; ++i) {
                        const double lo_n = UB ? lo_u : sLo[(i + 1) * 16 + j], hi_n = UB ? hi_u : sHi[(i + 1) * 16 + j];
                        if constexpr (SOC) { if (i + SPD < N) gr[(i + SPD) % SPR] = sC[cw + (i + SPD) * SLOT_D + PL_GC]; }
                        if constexpr (LSP) { if (i + SPD < N) glr[(i + SPD) % SPR] = sLG[cl + (i + SPD) * CSL]; }
                        if constexpr (LTP) { if (i + SPD < N) gtr[(i + SPD) % SPR] = sTG[cl + (i + SPD) * CSL]; }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (FUSED) {                  // first half of slot i's update in front of the chains (no s_nop)
                            double tt, vn, xn, t = Dn[i];
                            const double xi = X[i];
                            if constexpr (HALF) {
                                double gnew;
                                fused_forward_step_half<NX, NU>(tt, vn, t, xn, gnew, xi, G[i], lo_c, hi_c, mf1, mf2);
                                X[i + 1] = xn;
                                if constexpr (!LAZY_RES) {
                                    pmax = resid_max<true>(pmax, xi - vn);
                                    dmax = resid_max<true>(dmax, VP[i] - vn);
                                }
                                G[i] = gnew;
                                VN[i] = vn;
                                lo_c = lo_n; hi_c = hi_n;
                                continue;
                            }
                            fused_forward_step<NX, NU>(tt, vn, t, xn, xi, G[i], lo_c, hi_c, mf1, mf2);
                            X[i + 1] = xn;
                            if constexpr (!LAZY_RES) {
                                pmax = resid_max<true>(pmax, xi - vn);
                                dmax = resid_max<true>(dmax, VP[i] - vn);
                            }
                            G[i] = tt - vn;
                            VN[i] = vn;
                            if constexpr (SOC) sC[cw + i * SLOT_D] = fma(xi, socmask, gr[i % SPR]);   // x + gc -> cone step (below); see slot_update
                            lo_c = lo_n; hi_c = hi_n;
                            continue;
                        }
                        const double t = ring_sum<MODE, 0, NX>(Dn[i], X[i], mf1);   // f + A x_i | u_i = -d_i - Kinf x_i
                        X[i + 1] = ring_short<MODE, NX, NU>(t, t, mf2);             // x_{i+1} = (f + A x_i) + B u_i | u_i (slot i+1)
                        slot_update(i, lo_c, hi_c, gr[SOC ? i % SPR : 0], glr[LSP ? i % SPR : 0], gtr[LTP ? i % SPR : 0]);
                        lo_c = lo_n; hi_c = hi_n;
                    }
                    slot_update(N - 1, lo_c, hi_c, gr[SOC ? (N - 1) % SPR : 0], glr[LSP ? (N - 1) % SPR : 0], gtr[LTP ? (N - 1) % SPR : 0]);
                    if constexpr (LPL) {
                        // ---- half-space projections (admm.cpp:137-211) + their dual update (:239-254), transposed
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if constexpr (LSP) {
                            if (lin_x_on) project_halfspace_columns<NX, KMAX, 16, 16>(j, grp, N, CSL, 0, 0, sLV, sLG, sLin, 0, P.n_lin);
                            if (lin_u_on) project_halfspace_columns<NU, KMAX, 16, 16>(j, grp, N, CSL, NX, 1, sLV, sLG, sLin, 0, P.n_lin);
                        }
                        if constexpr (LTP) {
                            if (tlin_x_on) project_halfspace_columns<NX, KMAX, 16, 16>(j, grp, N, CSL, 0, 0, sTV, sTG, sTLin, 3 * KMAX * 16, P.n_tlin);
                            if (tlin_u_on) project_halfspace_columns<NU, KMAX, 16, 16>(j, grp, N, CSL, NX, 1, sTV, sTG, sTLin, 3 * KMAX * 16, P.n_tlin);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    // ---- termination_condition, admm.cpp:310-328 (the box residuals: the cone slacks do not enter them)
                    bool conv = false;
                    // The 2N subtractions and maxima behind the four residuals (a tenth of the quadrotor iteration) are only formed when the
                    // test can pass: a row converges only if EVERY entry of its residuals is below the tolerance, so two probe slots decide
                    // first -- slot 1 (x_1 | u_0: where the constraints bite) and the last one.  An entry there at or above its tolerance and
                    // the test is lost whatever the other slots hold (max >= that entry; `x rho` is monotone for rho > 0; a NaN entry is
                    // ignored by the maxima, so it does not count against the probe either).  The probe is skipped -- the full test runs --
                    // on the last test of the launch's iteration budget: the residuals it leaves are the ones the solve reports (:314-317).
                    auto termination = [&]() {
                        if (countdown > 0 && --countdown == 0) {
                            countdown = P.check_termination;
                            checked = 1;
                            constexpr int PS0 = N >= 2 ? 1 : 0, PS1 = N - 1;
                            auto lost = [&](const int s) {              // (`|`, not `||`: four compares and three s_or, no EXEC-mask regions)
                                return (int)(fabs(X[s] - VN[s]) >= P.tol_pri) | (int)(fabs(VP[s] - VN[s]) * rho >= P.tol_dua);
                            };
                            const bool maybe = ((int)!(rho > 0.0) | (int)!(lost(PS0) | lost(PS1))) != 0;
                            // any row all of whose lanes say maybe?  (scalar: AND-fold the ballot over each row's RL bits; rows that have
                            // left the loop are masked off and vote 0)
                            unsigned long long fold = __ballot(maybe);
                            fold &= fold >> 1; fold &= fold >> 2; fold &= fold >> 4;
                            if constexpr (!HALF) fold &= fold >> 8;
                            const bool any_row = (fold & (HALF ? 0x0101010101010101ull : 0x0001000100010001ull)) != 0ull;
                            // INVARIANT: rp / rd -- what d_resid reports and what repack_sort keys on -- are refreshed only by a
                            // FULL test: a row passed the probe, or this is the last test the launch's iteration budget allows.  Every launch
                            // and every stage of a split solve therefore ends on a full test BECAUSE its end is P.max_iter (the stage's cap);
                            // a caller that stopped a launch any other way would read residuals of an earlier iteration.
                            const bool last_test = (P.max_iter - 1 - it) < P.check_termination;
                            if (last_test || any_row) {
                                if constexpr (LAZY_RES) {
#pragma unroll
                                    for (int s = 0; s < N; ++s) {
                                        pmax = resid_max<false>(pmax, X[s] - VN[s]);
                                        dmax = resid_max<false>(dmax, VP[s] - VN[s]);
                                    }
                                }
                                rp = pmax;
                                rd = dmax * rho;
                                const bool ok = (rp < P.tol_pri) && (rd < P.tol_dua);
                                const unsigned long long bal = __ballot(ok);
                                conv = ((bal >> (grp * RL)) & RMASK) == RMASK;
                            }
                        }
                    };
                    if constexpr (SOC) {
                        // ---- cone step (admm.cpp:112-135, 228-235), transposed: x + gc of every slot went to LDS above; lane j
                        // of pass p gathers the three components of its (cone, knot) item, projects them -- ONE square root /
                        // division sequence per pass instead of one per knot -- and writes the item's cells of the three planes.
                        // The termination test needs nothing from it and stands between the first gather and its use.
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        double n0 = sC[item_at[0]], n1 = sC[item_at[0] + 1], n2 = sC[item_at[0] + 2];
                        // the next backward sweep's first terms: on their way together with the gather.  They stay valid unless a pass
                        // below takes its exact path and rewrites W cells (then they are read again)
#pragma unroll
                        for (int d = 1; d <= SPR && d <= N; ++d) wr[(N - d) % SPR] = sC[cw + (N - d) * SLOT_D];
                        termination();
                        bool any_exact = false;
#pragma unroll
                        for (int p = 0; p < SOC_PASSES; ++p) {
                            if (p > 0 && p >= soc_passes) break;            // wave-uniform
                            const int at = item_at[p];
                            if (p > 0) { n0 = sC[at]; n1 = sC[at + 1]; n2 = sC[at + 2]; }
                            const double s0 = n0, s1 = n1, s2 = n2;
                            if (soc_all_inside(s0, s1, s2, item_mu[p])) {
                                // no cone of the wave is active: vcnew = x + gc bit for bit, so gc = (x + gc) - vcnew = 0 and
                                // vcnew - gc = x + gc -- what the forward sweep left in the W plane.  VC takes its copy (for the
                                // write-back); the GC cells are zeroed once, then known to be zero (gcz: of the rows still iterating)
                                sC[at + PL_VC] = s0; sC[at + PL_VC + 1] = s1; sC[at + PL_VC + 2] = s2;
                                if (!gcz[p]) {
                                    sC[at + PL_GC] = 0.0; sC[at + PL_GC + 1] = 0.0; sC[at + PL_GC + 2] = 0.0;
                                    gcz[p] = true;
                                }
                            } else {
                                double r0, r1, r2;
                                soc_project3(s0, s1, s2, item_mu[p], item_rmu[p], mu_pow2, r0, r1, r2);
                                // the item's three cells of every plane (the dummy item: (0, 0, 1) onto itself, from every lane that has none)
                                const double g0 = s0 - r0, g1 = s1 - r1, g2 = s2 - r2;      // :229 / :234  (gc + x) - vcnew
                                sC[at + PL_VC] = r0; sC[at + PL_VC + 1] = r1; sC[at + PL_VC + 2] = r2;
                                sC[at + PL_GC] = g0; sC[at + PL_GC + 1] = g1; sC[at + PL_GC + 2] = g2;
                                sC[at] = r0 - g0; sC[at + 1] = r1 - g1; sC[at + 2] = r2 - g2;   // vcnew - gc: the next backward sweep's term
                                gcz[p] = false;
                                any_exact = true;
                            }
                        }
                        // a cell outside every item: gc = (x + gc) - vcnew = 0 from the solve's first iteration on (whatever the warm start
                        // held there); once per solve, wave-uniform (the rows of a wave open their solves together)
                        if (it == iter0 && !proj_lane) {
#pragma unroll
                            for (int s = 0; s < N; ++s) sC[cw + s * SLOT_D + PL_GC] = 0.0;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (any_exact) {
#pragma unroll
                            for (int d = 1; d <= SPR && d <= N; ++d) wr[(N - d) % SPR] = sC[cw + (N - d) * SLOT_D];
                        }
                    }
                    iter += 1;                                                      // :394
                    if constexpr (ADAPT) {
                        // ---- adaptive rho, admm.cpp:397-423: every 5th pass of the loop index (after iterations 6, 11, ...).
                        // benchmark_rho_adaptation (rho_benchmark.cpp:207-249) builds dense OSQP-style matrices and multiplies
                        // them out; every entry it multiplies by zero is skipped here, what is left is evaluated block by block:
                        //   rows of A x - z:  u_i - znew_i  and  (A x_i + B u_i - x_{i+1}) - vnew_{i+1}          (:78-102, :144-150)
                        //   P x + q + A' y:   (Q x_i | Pinf x_{N-1}) + Q x_i + (A' g_{i+1} - g_i),  R u_i + R u_i + (y_i + B' g_{i+1})   (:107-168)
                        if (it > 0 && it % 5 == 0) {
                            double at[NX];
#pragma unroll
                            for (int k = 0; k < NX; ++k) at[k] = sTab[j * AKC + k];
                            double pri_res = 0.0, ax_max = 0.0, z_max = 0.0, dual_res = 0.0, px_max = 0.0, aty_max = 0.0, q_max = 0.0;
                            double pxq_max = 0.0;              // max |Q x_i|, |R u_i| over the knots before the last: entries of P x AND of q
#pragma unroll
                            for (int i = 0; i < N - 1; ++i) {
                                // primal rows of knot i live in slot i+1: input lanes u_i, state lanes the dynamics defect
                                const double axs = ring_sum<MODE, 0, NX>(0.0, X[i], mf1);            // state lanes: A x_i
                                const double dyn = ring_short<MODE, NX, NU>(axs, is_input ? X[i + 1] : 0.0, mf2) - X[i + 1];
                                const double a = is_state ? dyn : X[i + 1];
                                const double zz = VN[i + 1];
                                const bool on = is_state || is_input;
                                pri_res = fmax(pri_res, on ? fabs(a - zz) : 0.0);
                                ax_max = fmax(ax_max, on ? fabs(a) : 0.0);
                                z_max = fmax(z_max, on ? fabs(zz) : 0.0);
                                // dual rows: state lanes knot i, input lanes knot i (slot i+1); A' g_{i+1} | B' g_{i+1} from one chain
                                const double init = is_state ? (i >= 1 ? -G[i] : 0.0) : G[i + 1];
                                const double aty = ring_sum<MODE, 0, NX>(init, G[i + 1], at);
                                const double xs = is_state ? X[i] : X[i + 1];
                                const double px = qr * xs;                                          // P block = Q | R (:114, :119)
                                dual_res = fmax(dual_res, on ? fabs((px + px) + aty) : 0.0);       // q = Q x | R u (:132, :137)
                                pxq_max = fmax(pxq_max, on ? fabs(px) : 0.0);
                                aty_max = fmax(aty_max, on ? fabs(aty) : 0.0);
                            }
                            {   // last knot (state lanes): P block = the CURRENT Pinf (:112), A' y = -g_{N-1}
                                double prow[NX];
#pragma unroll
                                for (int k = 0; k < NX; ++k) prow[k] = sP[grp * NX * NX + (is_state ? j : 0) + NX * k];   // Pinf[j][k]
                                const double pxl = ring_sum<MODE, 0, NX>(0.0, X[N - 1], prow);
                                const double ql = qr * X[N - 1];
                                const double atyl = -G[N - 1];
                                dual_res = fmax(dual_res, is_state ? fabs((pxl + ql) + atyl) : 0.0);
                                px_max = fmax(pxq_max, is_state ? fabs(pxl) : 0.0);
                                aty_max = fmax(aty_max, is_state ? fabs(atyl) : 0.0);
                                q_max = fmax(pxq_max, is_state ? fabs(ql) : 0.0);                  // q uses Q at the last knot too (:132)
                            }
                            pri_res = grp_max16(pri_res); ax_max = grp_max16(ax_max); z_max = grp_max16(z_max);
                            dual_res = grp_max16(dual_res); px_max = grp_max16(px_max); aty_max = grp_max16(aty_max); q_max = grp_max16(q_max);
                            // predict_rho, rho_benchmark.cpp:174-194
                            const double pri_norm = fmax(ax_max, z_max);
                            const double dual_norm = fmax(fmax(px_max, aty_max), q_max);
                            const double eps = 1e-10;
                            const double normalized_pri = pri_res / (pri_norm + eps);
                            const double normalized_dual = dual_res / (dual_norm + eps);
                            const double ratio = normalized_pri / (normalized_dual + eps);
                            double new_rho = rho * sqrt(ratio);
                            if (P.aclip) new_rho = fmin(fmax(new_rho, P.arho_min), P.arho_max);
                            // update_matrices_with_derivatives, rho_benchmark.cpp:196-210 (the second call at admm.cpp:421 adds 0)
                            const double delta = new_rho - rho;
                            if (is_state) {
#pragma unroll
                                for (int k = 0; k < NU; ++k) mb[NX + k] = -taylor_step(-mb[NX + k], delta, sTab[(16 + j) * AKC + k]);
#pragma unroll
                                for (int k = 0; k < NX; ++k) {
                                    const int e = grp * NX * NX + k + NX * j;
                                    sP[e] = taylor_step(sP[e], delta, sTab[(32 + j) * AKC + k]);
                                }
                            } else if (is_input) {
#pragma unroll
                                for (int k = 0; k < NX; ++k) mf1[k] = -taylor_step(-mf1[k], delta, sTab[(16 + j) * AKC + k]);
                            }
                            if (P.aC1 || P.aC2) {                                           // C1 / C2: logged, applied by flush_c
                                if (j == 0) sDl[grp * ADAPT_LOG + ndl] = delta;
                                if (++ndl == ADAPT_LOG) { flush_c(ndl); ndl = 0; }
                            }
                            rho = new_rho;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            // p[:,N-1] = -(Xref[:,N-1]' Pinf) - ... is evaluated with the Pinf of the moment (admm.cpp:292)
                            QX[N - 1] = qx_last_plain;
                            terminal_term();
                        }
                    }
                    if constexpr (!SOC) termination();                             // (after the adaptation: it may have moved rho)
                    if (conv) { solved = 1; break; }                                // :431-441 (returns before v = vnew)
#pragma unroll
                    for (int s = 0; s < N; ++s) VP[s] = VN[s];                      // :445-446
                    vp_touched = true;
                }
                acc_iter += (unsigned)(iter - iter0);
                acc_solved += (unsigned)solved;
                // (the logs of a single-step launch: a stretch of a fused launch that batch_dispatch.hip cut up, "step_regroup")
                if (P.iter_log && j == 0) P.iter_log[(size_t)step * P.batch + b] = solved ? iter : -iter;
                if (P.u0_log && is_input) P.u0_log[((size_t)step * P.batch + b) * NU + (j - NX)] = X[1];
                if (nsteps > 1) {
                    // plant step x0 <- A x0 + B u_0 + f (== forward_pass x_1).  Input lanes hold u_0 in slot 1: their
                    // slot 0 is the neutral dummy and must stay 0, or its "slack" would enter the dual residual
                    x0v = is_state ? X[1] : 0.0;
                }
            }

            // ---- write back (coalesced) -------------------------------------------------------
            // FLS: every store instruction writes WHOLE knot segments [x_s ; u_s] (128 bytes = one cache line at nx+nu = 16).  The slot
            // convention -- input lanes hold knot s-1 at slot s -- otherwise makes a store write 96 bytes of one line and 32 of the line
            // before it: partial-line stores, and HBM takes those at three quarters of the rate of whole lines (tools/ubench/
            // ubench_stream_forms.hip -DSLOT_MAJOR=1 -DSPLIT_LINES=1: the copy pattern of a warm solve 5.7 -> 4.4 TB/s, and the
            // PREFETCH form turns from 20 % faster than one tile per wave into 7 % slower).  So the input lanes store the value of their
            // NEXT slot into the segment of the state lanes' slot; the input cells of the last knot (no u_{N-1}: never read, zero since
            // the records were allocated) are written as zero.  Same values at the same addresses: bit-identical records.
            constexpr bool FLS = TINYMPC_FULL_LINE_STORES != 0 && (PF || TINYMPC_FULL_LINE_STORES == 2) && !SOC && LIN == 0 && !DBG;
            if (FLS && (P.store_mask & 32) == 0) {
                if constexpr (FLS) {
                    const size_t seg = (size_t)b * (N * NZ) + j;
                    const bool in_row = j < NZ;
#pragma unroll
                    for (int s = 0; s < N; ++s) {
                        const size_t off = seg + s * NZ;
                        const double xs = is_input ? (s + 1 < N ? X[s + 1 < N ? s + 1 : s] : 0.0) : X[s];
                        const double vs = is_input ? (s + 1 < N ? VN[s + 1 < N ? s + 1 : s] : 0.0) : VN[s];
                        const double gs = is_input ? (s + 1 < N ? G[s + 1 < N ? s + 1 : s] : 0.0) : G[s];
                        const double ps = is_input ? (s + 1 < N ? VP[s + 1 < N ? s + 1 : s] : 0.0) : VP[s];
                        if (in_row) {
                            if (P.store_mask & 1) {
                                // max_iter = 0: the sweeps never ran, x[:,1:] and u keep what they held (only x[:,0] = x0 is set)
                                if (acc_iter > 0) { if constexpr (PF) P.prim[off] = xs; else store_primal(P.prim + off, xs); }
                                else if (s == 0 && is_state) P.prim[off] = xs;
                            }
                            if (P.store_mask & 2) P.slack[off] = vs;
                            if (P.store_mask & 4) P.dual[off] = gs;
                            if ((P.store_mask & 8) && vp_touched) P.slack_prev[off] = ps;   // admm.cpp:431-441 returns before v = vnew
                        }
                    }
                }
            } else
#pragma unroll
            for (int s = 0; s < N; ++s) {
                const bool valid = is_state || (is_input && s >= 1);
                const size_t off = lbase + s * NZ;
                if (valid) {
                    // max_iter = 0: the sweeps never ran, x[:,1:] and u keep what they held (only x[:,0] = x0 is set)
                    // bit 0: the whole x|u trajectory; bit 5: only its first knot (x_0, x_1, u_0: what a closed-loop caller applies)
                    if (((P.store_mask & 1) || ((P.store_mask & 32) && s <= 1)) && (acc_iter > 0 || (s == 0 && is_state))) {
                        // (a cone / half-space variant reads x|u back at the start of the next solve, admm.cpp:352-374: plain store)
                        if constexpr (SOC || LIN != 0) P.prim[off] = X[s];
                        else store_primal(P.prim + off, X[s]);
                    }
                    if (P.store_mask & 2) P.slack[off] = VN[s];
                    if (P.store_mask & 4) P.dual[off] = G[s];
                    if ((P.store_mask & 8) && vp_touched) P.slack_prev[off] = VP[s];   // admm.cpp:431-441 returns before v = vnew
                    if constexpr (SOC) {
                        // a family whose cone switch is off keeps its records (admm.cpp:102-109, 228-235).  vcnew: the VC plane where the
                        // cell belongs to an item (or no iteration ran: what the solve started from), else what the last forward sweep
                        // left in the W plane
                        if (soc_lane && (P.store_mask & 16)) {
                            P.cslack[off] = sC[cw + s * SLOT_D + ((proj_lane || iter <= iter_first) ? PL_VC : 0)];
                            P.cdual[off] = sC[cw + s * SLOT_D + PL_GC];
                        }
                    }
                    if constexpr (LSP) { if (lin_lane && (P.store_mask & 16)) { P.lslack[off] = sLV[cl + s * CSL]; P.ldual[off] = sLG[cl + s * CSL]; } }
                    if constexpr (LTP) { if (tlin_lane && (P.store_mask & 16)) { P.tlslack[off] = sTV[cl + s * CSL]; P.tldual[off] = sTG[cl + s * CSL]; } }
                    if constexpr (LSR) { if (lin_lane && (P.store_mask & 16)) { P.lslack[off] = VL[s]; P.ldual[off] = GL[s]; } }
                    if constexpr (LTR) { if (tlin_lane && (P.store_mask & 16)) { P.tlslack[off] = VT[s]; P.tldual[off] = GT[s]; } }
                    if constexpr (DBG) {
                        if (P.dbg_qr && acc_iter > 0) {
                            P.dbg_qr[off] = Qd[s];                          // work->q | work->r
                            P.dbg_pd[off] = is_state ? Pd[s] : Dd[s];       // work->p | work->d
                        }
                    }
                }
            }
            if constexpr (ADAPT) {
                if (ndl > 0) flush_c(ndl);
                if (j == 0) P.arho[b] = rho;
                if (is_state) {
#pragma unroll
                    for (int k = 0; k < NU; ++k) P.aK[(size_t)b * (NU * NX) + k + NU * j] = -mb[NX + k];
#pragma unroll
                    for (int k = 0; k < NX; ++k) P.aP[(size_t)b * (NX * NX) + k + NX * j] = sP[grp * NX * NX + k + NX * j];
                }
            }
            if (P.x0_next && acc_iter > 0 && is_state) P.x0_next[(size_t)b * NX + j] = X[1];     // x1 = A x0 + B u0 + f (needs a forward pass)
            const double ps = grp_maxw<RL>(is_state ? rp : 0.0), pi = grp_maxw<RL>(is_input ? rp : 0.0);
            const double ds = grp_maxw<RL>(is_state ? rd : 0.0), di = grp_maxw<RL>(is_input ? rd : 0.0);
            if (P.next_index) {
                const bool open = j == 0 && !solved;
                const unsigned long long m = __ballot(open);
                if (m) {
                    const int first = __ffsll((long long)m) - 1;
                    int at = 0;
                    if (lane == first) at = atomicAdd(P.next_count, __popcll(m));
                    at = __shfl(at, first);
                    if (open) P.next_index[at + __popcll(m & ((1ull << lane) - 1ull))] = b;
                }
            }
            if (j == 0) {
                P.status[b] = make_int4(iter, solved, solved ? 1 : 11, checked);
                double4 rr = make_double4(ps, pi, ds, di);
                *reinterpret_cast<double4*>(P.resid + (size_t)b * 4) = rr;
                if (P.accum) {
                    if constexpr (PF) {                      // (no value comes back: a persistent wave must not wait for one here)
                        __hip_atomic_fetch_add(&P.accum[b].x, acc_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(&P.accum[b].y, acc_solved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        uint2 ac = P.accum[b];
                        ac.x += acc_iter;
                        ac.y += acc_solved;
                        P.accum[b] = ac;
                    }
                }
            }
        }
    }
}

}  // namespace tinympc_amd
