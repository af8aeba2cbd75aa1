// batch_impl.hpp -- the opaque TinyBatch behind include/tinympc_amd.h (shared by batch_api.hip, batch_dispatch.hip, batch_tables.hip, batch_helpers.hip and
// compat_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/tinympc_amd.h"
#include "kernel_entry.hpp"
#include "general_kernel.hip.h"
#include "jit.hpp"
#include "riccati_kernel.hip.h"
#include "tile_kernel.hip.h"
#include "cache.hpp"

namespace tinympc_amd {

struct Settings {              // TinySettings (types.hpp:63-82) hot-path subset; defaults tiny_api.cpp:413-441
    double abs_pri_tol = 1e-3, abs_dua_tol = 1e-3;
    int max_iter = 1000, check_termination = 1;
    int en_state_bound = 1, en_input_bound = 1, en_state_soc = 0, en_input_soc = 0;
    int en_state_linear = 0, en_input_linear = 0, en_tv_state_linear = 0, en_tv_input_linear = 0;
};

}  // namespace tinympc_amd

constexpr size_t KPI_SKEW_BYTES = 0;             // stagger between the starts of the record arrays inside their slab (see tiny_batch_setup)
struct TinyBatch {
    int nx = 0, nu = 0, N = 0, batch = 0, device = 0, num_cus = 256;
    const tinympc_amd::KernelEntry* kernel = nullptr;
    const tinympc_amd::TileEntry* tile = nullptr;      // tile_kernel.hip.h instantiation for this shape, if any
    double* d_ttab = nullptr;
    size_t ttab_doubles = 0;
    std::vector<double> h_ttab;
    tinympc_amd::TileEntry tile_dyn = {0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr};   // tile shape chosen at run time (b->tile points here; jit.hpp)
    bool tile_is_jit = false, tile_soc_failed = false;
    bool redispatch = false;                     // launch_solve re-entered by itself after a failed instantiation (not a caller-side change)
    bool no_jit = false, jit_failed = false, variant_jit_failed = false;     // run-time instantiation of the one-row kernel for shapes outside kernel_dims.txt (jit.hpp)
    int tile_verdict = 0, tile_since = 0;        // the dynamic tile form tried on a one-row shape: 1 kept, -1 rejected, 0 open (batch_dispatch.hip launch_solve)
    double tile_rate = 0.0;
    bool probe_was_tile = false;
    int tile_w = -1;                             // option "tile_w"
    int tile_lm = -1;                            // option "tile_lm"
    int tile_dyn_opt = -1;                       // option "tile_dyn"
    bool het_ub = true;                          // option "het_ub": the per-instance-data variant's UB form (run-time instantiated) where the box allows it
    bool last_tile_dyn = false;
    int last_tile_form = -1;                     // W * 1e6 + R * 1e3 + LM of the entry the last tile launch took
    int* d_work_counter = nullptr;               // the dynamic tile form's device-wide instance counter
    int tile_r = 0;                              // option "tile_r": pick the tile_dims.txt entry with this many rows along the horizon
    bool no_tile = false, prefer_tile = false;   // prefer_tile: take the tile kernel even where a one-row instantiation exists
    // host copies of the problem family
    tinympc_amd::Mat A, B, f;
    std::vector<double> Qw, Rw;                    // work->Q, work->R (user + rho)
    tinympc_amd::Cache cache;
    tinympc_amd::Settings set;
    bool have_bounds = false;
    std::vector<double> x_min, x_max, u_min, u_max;
    std::vector<int> Acx, qcx, Acu, qcu;
    std::vector<double> cx, cu;
    bool cones_overlap_x = false, cones_overlap_u = false;   // cones of a family share rows (admm.cpp:111-135 projects them one after the other): coverage kernel
    // half-space constraints a_k' z <= b_k (row-major copies: [k][n]); time-varying: [knot][k][n]
    int nsl = 0, nil = 0, ntsl = 0, ntil = 0;
    std::vector<double> Alin_x, blin_x, Alin_u, blin_u, tvA_x, tvb_x, tvA_u, tvb_u;
    // device state (KPI records, see admm_kernel.hip.h)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t d_tab_doubles = 0;                    // capacity of d_tab (grows with the half-space table stride)
    double *d_tab = nullptr, *d_x0 = nullptr, *d_ref = nullptr, *d_prim = nullptr, *d_slack = nullptr,
           *d_dual = nullptr, *d_slack_prev = nullptr, *d_cslack = nullptr, *d_cdual = nullptr, *d_resid = nullptr,
           *d_stage = nullptr, *d_stats = nullptr, *d_dbg_qr = nullptr, *d_dbg_pd = nullptr;
    void* d_kpi_slab = nullptr;                  // the allocation behind d_ref ... d_cdual (batch_api.hip: skewed starts)
    int4* d_status = nullptr;
    uint2* d_accum = nullptr;
    double *d_lslack = nullptr, *d_ldual = nullptr, *d_tlslack = nullptr, *d_tldual = nullptr, *d_gtab = nullptr;
    size_t gtab_doubles = 0, general_lds_limit = 0, phase_lds_limit = 0;
    std::vector<double> h_gtab;
    tinympc_amd::GeneralArgs gargs;      // table offsets filled by build_general_tables
    bool force_general = false;
    int* d_iter_log = nullptr;
    double* d_u0_log = nullptr;
    int log_steps = 0;
    size_t stage_doubles = 0;
    std::vector<double> h_tab;
    bool tab_dirty = true;
    unsigned tab_gen = 1, ttab_gen = 0;          // generation of the problem data (bumped by the first launch that sees tab_dirty) / the one h_ttab was built from
    int last_path = -1;
    // options
    bool advance_x0 = false, debug = false;
    int grid_waves_per_cu = 0, dpp_mode = 2, steps_per_launch = 1;
    bool step_log = false, reset_duals = false;
    // repack_after = K > 0: a solve is split at iterations K, 2K, 4K, ... -- the instances that have not converged by then are
    // compacted and carried on by the next launch with four of them per wave again (divergent cold batches: a slow
    // instance no longer holds a wave by itself).  Results are bit-identical to the unsplit solve.
    int repack_after = -1;               // K > 0: split at K; 0: never; -1 (default): K picked from the previous solve's iteration histogram
    // automatic split: histogram of the per-instance iteration counts of the last eligible solve (device -> pinned host,
    // asynchronous), and the K the cost model derived from it (0 = a plain launch is predicted to be as fast)
    enum { HIST_BINS = 1024 };
    unsigned *d_hist = nullptr, *h_hist = nullptr;
    hipEvent_t hist_ev = nullptr;
    bool hist_pending = false;
    int auto_cap = 0, auto_cap_max_iter = 0;
    // the model's proposal is then checked against the clock: time per instance-iteration of the last plain and the last split
    // launch (HIP events around the launches of an eligible solve, read when the histogram arrives)
    hipEvent_t auto_ev0 = nullptr, auto_ev1 = nullptr;
    int auto_last_cap = 0;               // the cap the timed launch ran with
    int auto_probes = 0;                 // timed solves so far (the first one carries one-time costs -- code object load -- and is not used)
    double auto_plain_rate = 0.0, auto_split_rate = 0.0;
    int auto_since = 0;                  // eligible solves since the last verdict: every 32nd one re-opens the question (plain + split timed again)
    int auto_verdict = 0;                // 0 undecided, 1 the split was measured faster (kept), -1 measured slower (plain launches from now on)
    double auto_gain = 0.0;              // predicted time of the split solve / plain solve (diagnostics)
    bool repack_dynamic = false;                 // follow-up stages of a split solve take their tiles off a per-stage counter (measured on config 3: 1-2 % SLOWER than the grid stride)
    int repack_waves_per_cu = 8, repack_growth = 0;   // grid of the follow-up stages; stage s runs to K * growth^s (0: the cost model picks 2 or 4 with K)
    int auto_growth = 2, growth_alt = 2, growth_verdict = 0;   // the stage schedule in use; the one on trial; 1 = the clock has compared the two
    bool probe_was_growth = false;
    // repack_sort: the list of open instances a stage of a split solve takes is ordered by how far each is from its tolerances
    // (largest residual / tolerance ratio first; 16 bins per octave, a counting sort between the stages): ADMM converges at a roughly
    // geometric rate, so rows that are equally far out leave their wave together.  1: every follow-up stage; 0: never; -1 (default):
    // the stages that the last iteration histogram predicts to run for at least 60 us
    int repack_sort = -1;
    std::vector<unsigned> hist_copy;     // the histogram the schedule in use was derived from (h_hist is overwritten asynchronously)
    int last_sorted_stages = 0;
    int *d_repack_index = nullptr, *d_repack_count = nullptr;
    bool use_ub = true;                            // option "uniform_bounds": take the UB kernel variant when the box allows it
    bool bounds_uniform = false;                   // build_tables: every knot has the same box (admm_kernel.hip.h UB variant)
    bool tile_bounds_uniform = false;              // build_tile_tables_w: the same for the tile kernel's UB form
    bool xref_shared = true, uref_shared = true;   // the Xref / Uref records of all instances are identical (broadcast, or still zero)
    bool share_ref = true;                         // option "share_ref": let launches exploit that
    int half_rows = -1;                  // option "half_rows": the one-row kernel's HALF form (nx+nu <= 8, eight instances per wave) where it exists; 0 = off
    bool last_half = false;              // the last one-row launch took it
    int launch_order = 1;                // option "launch_order": 1 = successive plain launches of the one-row kernel walk the batch in alternating directions
    bool order_flip = false;
    // PREFETCH form of the one-row kernel (round 6; admm_kernel.hip.h PF): persistent waves that draw their tiles from a ticket counter
    // and whose NEXT tile's records travel into the wave's LDS buffer (LDS-DMA) while the current one iterates.  Option "prefetch":
    // -1 (default) every plain single-step launch of a batch of at least PF_AUTO_MIN_TILES tiles per resident wave and the first stage
    // of a split solve; 0 never; 1 wherever the form exists (any batch size: tests).  "prefetch_waves": cap on the persistent grid
    // (0: what is resident)
    int prefetch = -1, prefetch_waves = 0;
    int prefetch_static = -1;            // option "prefetch_static": percent of a wave's tiles it takes by grid stride (the rest by ticket); -1: by
                                         // rule -- 75 for warm launches (tiles cost alike: profiles/r06_prefetch_probe.md), 50 for cold ones (a split
                                         // solve's first stage: tiles differ by what their rows need; config 3 0.762 -> 0.749 ms, tools/experiments/config3_knobs.py)
    unsigned* d_pf_counter = nullptr;    // eight ticket counters (64 bytes apart), never reset: a launch draws a known number from each
    unsigned pf_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // where the next launch's tickets begin, per shard
    bool last_prefetch = false;          // the last one-row launch (a split solve: its first stage) took the form
    // shipped plans (round 6): the settled TinyBatchPlans of the BASELINE shapes travel with the library (tinympc_amd/data/plans.txt); a
    // fresh handle whose shape, settings and batch bucket match one takes its launch form on the FIRST solve instead of probing.
    // Option "plan" = 0: off.  Read-back "plan_shipped"
    int plan_opt = 1;
    // layout of d_het_tabs as tiny_batch_setup_hetero built it (columns x lanes per column: 16 x 16 one-row kernel, 32 x 16 W tile kernel);
    // a launch whose kernel reads another layout is refused instead of reading the tables with the wrong stride
    int het_tab_cols = 0, het_tab_lw = 0;
    // one_shot on the tile kernel (round 6): option "one_shot_fast" = 1 (default) lets a one-shot launch ride on the shape's fast box
    // form -- LDS-offload set, v|z streamed to d_vz_scratch instead of its record, dynamic slots --, 0 keeps the all-in-registers form
    bool one_shot_fast = true;
    double* d_vz_scratch = nullptr;
    // the split solve's TAIL on the tile kernel (round 6): after the capped first stage the open instances -- listed by that launch -- run
    // to max_iter in ONE launch of the shape's dynamic slot form (tile_dims.txt: its one-row layout), whose rows refill one by one: no
    // follow-up stages in lock step.  Option "repack_tail": -1 by the clock (probed like the split itself), 0 never, 1 wherever the form
    // exists.  launch_tile reads tail_index / tail_count / tail_iter_base while enqueue_split_solve has them set
    int repack_tail = -1, tail_verdict = 0;
    double tail_rate = 0.0;
    const int* tail_index = nullptr;
    const int* tail_count = nullptr;
    int tail_iter_base = 0;
    bool last_tail_tile = false;
    const void* loaded_kernels[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // kernels whose first launch this handle has paid for
    bool helpers_loaded = false;          // batch_helpers.hip preload_helper_kernels
    bool plan_tried = false, plan_shipped = false;
    int last_pf_grid = 0;
    size_t last_pf_lds = 0;
    const void* pf_occ_kernel = nullptr; // residency of the form's kernel at pf_occ_lds bytes of dynamic LDS (asked once)
    size_t pf_occ_lds = 0;
    int pf_occ = 0;
    // step_regroup (fused closed-loop launches of the one-row kernel): the launch is cut into stretches of K MPC steps and every
    // stretch takes the instances ordered by the iteration count of their last solve (SolveArgs::perm; a counting sort on the
    // device between the stretches) -- the four rows of a wave run in lock step, and what a row needed at its last step says what
    // it will need at the next ones.  K > 0: stretches of K steps; 0: never; -1 (default): on when the iteration totals of the
    // previous fused launch of this batch (d_accum, grouped four by four as the waves take them) say that lock step costs >= 5 %.
    int step_regroup = -1;
    int regroup_verdict = 0, regroup_since = 0;    // 0 open, 1 on, -1 off (the estimate of the last fused launch decided)
    bool status_valid = false;                     // d_status holds this episode's last iteration counts (not after setup / reset)
    int* d_perm = nullptr;
    // option "step_regroup_streams": 2 (default) = the batch in two halves on two streams, their stretches half a stretch apart (a
    // half that drains at the end of a stretch leaves its CUs to the other half, which is in the middle of one): config 4 27.0 ms
    // against 27.4 on one stream; joined back into the batch's stream before the solve call returns.  (A second stream in the process
    // costs the launch-bound paths nothing: tools/second_stream_probe.py.)
    int regroup_streams = 2;
    hipStream_t stream2 = nullptr;
    hipEvent_t rg_fork = nullptr, rg_join = nullptr;
    unsigned* d_rg_bins = nullptr;                 // 1024 bins of the counting sort
    unsigned long long *d_ls = nullptr, *h_ls = nullptr;   // {4 x sum over waves of the largest total, sum of the totals}
    hipEvent_t ls_ev = nullptr;
    bool ls_pending = false;
    double lockstep_ratio = 0.0;                   // the last estimate (1.0 = rows of a wave agree)
    int last_regroup_stretches = 0;                // launches the last fused solve was cut into (1: not cut)
    int store_primal = 1;                // false: launches do not write x|u back (no consumer between closed-loop steps when the plant step runs on the device)
    bool records_zero = false, auto_cold = true; // every warm-start record is known to be zero (after tiny_batch_reset / setup): launches skip reading them
    int one_shot = 0;                    // 1: cold state assumed, x|u + vnew|znew written; 2: x|u only (bytes_cold of SURVEY.md 8(d))
    double* d_traj = nullptr;
    // heterogeneous problem families: per-instance problem data, caches and lane tables (device)
    bool hetero = false;
    double *d_hA = nullptr, *d_hB = nullptr, *d_hf = nullptr, *d_hQw = nullptr, *d_hRw = nullptr, *d_hrho = nullptr,
           *d_hK = nullptr, *d_hP = nullptr, *d_hQuu = nullptr, *d_hAmBKt = nullptr, *d_hAPf = nullptr, *d_hBPf = nullptr,
           *d_het_tabs = nullptr;
    int* d_hiters = nullptr;
    int* d_traj_offsets = nullptr;
    int traj_points = 0;
    long traj_step = 0;
    // adaptive rho (admm.cpp:397-423, rho_benchmark.cpp): settings, sensitivity tables (column-major), per-instance cache state
    bool adaptive = false, adaptive_clip = true, atab_dirty = true, astate_fresh = false;
    double adaptive_min = 1.0, adaptive_max = 100.0;
    std::vector<double> dKinf, dPinf, dC1, dC2;
    double *d_arho = nullptr, *d_aK = nullptr, *d_aP = nullptr, *d_aC1 = nullptr, *d_aC2 = nullptr, *d_atab = nullptr;
    // tiny_batch_allreduce_stats (group_api.hip): the gather table of the 64-byte statistics messages, device + pinned host
    double *d_wire = nullptr, *h_wire = nullptr;
    int wire_ranks = 0;
    // timing
    std::vector<hipEvent_t> ev_start, ev_stop;
    int timing_n = 0, timing_left = 0;
    char err[256] = {0};
};

namespace tinympc_amd {
int fail(TinyBatch* b, int code, const char* fmt, ...);
int launch_solve(TinyBatch* b);
bool apply_shipped_plan(TinyBatch* b);               // batch_api.hip: the matching entry of data/plans.txt, if any, imported into a fresh handle
int xfer_fields(TinyBatch* b, const TinyField* fields, const size_t* offsets, int n, double* d_buf, bool to_device,
                bool with_status, size_t off_status, size_t off_resid);
// project_soc (which = 0) / project_hyperplane (1) of an n-vector in device memory, one GPU thread, synchronous
int launch_projection(int which, double* v, const double* a, int n, float mu, double b);
}  // namespace tinympc_amd
