// tile_kernel.hip.h -- register-resident ADMM solve for shapes the one-row kernel (admm_kernel.hip.h) cannot
// hold: knot vectors wider than 16 rows and/or horizons whose per-lane arrays exceed one lane's registers.
//
// An instance occupies a TILE of W x R DPP rows of a wavefront (W * R <= 4):
//   * W = ceil((nx+nu)/16) rows ACROSS the stacked knot vector: lane (wrow, j16) owns row jj = 16*wrow + j16.
//     y = M w then needs the other row's half of w: one v_permlane16_swap_b32 pair turns w into
//     (a, b) = (even row's 16 values, odd row's 16 values) replicated in both rows, and the FMA chain is
//     sum_k bcast(a,k) M[jj][k] + sum_k bcast(b,k) M[jj][16+k] with the same v_fmac_f64_dpp row_newbcast blocks.
//   * R rows ALONG the horizon: row hrow owns slots [hrow*L, (hrow+1)*L), L = N/R, as L-long register arrays.
//     The Riccati sweeps are sequential in the knot index, so the rows take turns (EXEC masks the others) and
//     hand the running p_{i+1} / r_i (backward) or x_{i+1} | u_i (forward) to the neighbouring row through one
//     cross-row shuffle per phase; every element-wise phase runs on all rows at once.
// A whole wavefront's 128 KB of registers can therefore hold ONE large instance (e.g. nx=20, nu=8, N=50:
// W=2, R=2) or two / four smaller ones.  Same arithmetic, slot convention (input lanes keep knot i in slot i+1),
// HBM records and parity tests as the one-row kernel; box constraints, fused closed-loop MPC steps.
#pragma once
#include "admm_kernel.hip.h"

namespace tinympc_amd {

// table layout of the tile kernel (doubles): matrices [column k (0..31)][LW lanes], LW = 16*W
template <int W>
struct TileTab {
    static constexpr int LW = 16 * W;
    static constexpr int MB = 0, MF1 = 32 * LW, MF2 = 64 * LW, PT = 96 * LW, VEC = 128 * LW, BOUNDS = 128 * LW + 16 * LW;
    // half-space tables of the LIN variants behind the bounds: static [3][kmax][LW], then per slot [N][3][kmax][LW]
    static constexpr int lin_offset(int N) { return BOUNDS + 2 * N * LW; }
    static constexpr int tlin_offset(int N, int kmax) { return lin_offset(N) + 3 * kmax * LW; }
    static constexpr int doubles(int N, int kmax = LIN_KMAX) { return tlin_offset(N, kmax) + 3 * N * kmax * LW; }
};

// (the half-row FMA chains -- THL / THH, ring1_half -- live in admm_kernel.hip.h: the one-row kernel's HALF variant shares them)
// (a, b) = (values of the even DPP row, values of the odd DPP row) of each 32-lane half, visible in both rows
__device__ __forceinline__ void swap16(double v, double& a, double& b) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double(r1[0], r0[0]);
    b = __hiloint2double(r1[1], r0[1]);
}

// W = 2 on TWO accumulators (round 6, measured and NOT the default: -DTINYMPC_TILE_W2_CHAINS=2): each half of a two-row mat-vec (a:
// columns 0-15, b: 16-31) on the one-row kernel's two-chain block (admm_kernel.hip.h ring<0>: even columns on a0, odd ones on a1).  A
// lone wave per SIMD issues a DEPENDENT v_fmac_f64_dpp every 5.25 cycles and an independent one every 4.38
// (profiles/r01_ubench_fp64_dpp.txt), so a 24-column chain would save ~20 cycles -- and costs a zeroed second accumulator and a final
// add (two instructions = 8 cycles), plus what the allocator does with two more live registers at 512.  A/B of two libraries on one
// box, all twelve wide sweep cells, 131 072 instances: 3-9 % SLOWER ((20,4,50) 217.4 -> 237.1 ms, (20,8,10) 16.85 -> 18.34);
// iteration counts unchanged.  profiles/r06_negative_results.md.
#ifndef TINYMPC_TILE_W2_CHAINS
#define TINYMPC_TILE_W2_CHAINS 1
#endif

// init + sum_{k=C0}^{C1-1} M[jj][k] * w[k], w spread one entry per lane over W rows; m[k - C0] = M[jj][k]
template <int W, int C0, int C1>
__device__ __forceinline__ double tile_matvec(double init, double src, const double* m) {
    double acc = init;
    if constexpr (C1 > C0) {
        if constexpr (W == 0) {
            ring1_half<C0, C1 - C0>(acc, src, m);
        } else if constexpr (W == 1) {
            ring1<C0, C1 - C0>(acc, src, m);
        } else {
            double a, b;
            swap16(src, a, b);
            constexpr int E0 = C1 < 16 ? C1 : 16;          // end of the part that lives in the even row
            constexpr int S1 = C0 > 16 ? C0 : 16;          // start of the part that lives in the odd row
            if constexpr (TINYMPC_TILE_W2_CHAINS == 2 && C1 - C0 >= 4) {
                double a0 = init, a1 = 0.0;
                if constexpr (C0 < 16) ring<0, C0, E0 - C0>(a0, a1, a, m);
                if constexpr (C1 > 16) ring<0, S1 - 16, C1 - S1>(a0, a1, b, m + (S1 - C0));
                return a0 + a1;
            }
            if constexpr (C0 < 16) ring1<C0, E0 - C0>(acc, a, m);
            if constexpr (C1 > 16) ring1<S1 - 16, C1 - S1>(acc, b, m + (S1 - C0));
        }
    }
    return acc;
}

// LM (round 3): which per-lane arrays leave the register file.  bit 0: QX (the linear-cost constants, written once per solve, read
// once per backward step); bit 1: Dn (-d | fdyn, written by a backward step, read by the forward step of the same slot); bit 2: the
// x|u trajectory is not kept at all -- the sweep only needs its rolling value, and ONE extra forward pass after the last
// iteration reproduces it bit for bit from the d of that iteration (only where nothing reads x|u between solves: no cone,
// no half-spaces).  What stays in registers: g|y, vnew|znew, v|z -- 3 L-long arrays instead of 5 + the LDS trajectory.  That is
// what lets a lane hold a 50-knot horizon (R = 1: every row sweeps its own instance, no lanes idle in the sweeps) where round 2
// had to split the horizon over two rows.  The LDS arrays are COMPACT: [slot][row group][NZ] + one dummy entry per slot that all
// the lanes beyond nx+nu share (they only ever hold zeros).
// bit 4 (VPG, instead of bit 3): v|z is not held at all.  Inside the iteration loop it is only ever the vnew|znew of the iteration
// before -- which the slot update still has in its register when it needs it for the dual residual -- EXCEPT in a solve's first
// iteration (work->v of the solve before: admm.cpp:431-441 returns before v = vnew).  So: the slot update streams the old
// vnew|znew to the instance's v|z RECORD (one global store per slot; ~1 TB/s of HBM writes at (20,8,50), far from a bound), a
// solve's first iteration reads the record back into the dead vnew|znew registers behind its backward sweep, and a solve that
// ends without converging writes vnew|znew over it (v = vnew, :445-446).  Frees the largest LDS array of the long wide shapes:
// (20,8,50) 46 -> 23 KB per wave, four waves per CU instead of three.
// bit 5 (QXR, instead of bit 0): QX is not held either -- it is -(ref * Q|R diagonal) of a record that does not change during a solve, so
// the backward sweep reads the reference record again, TILE_QXR_DEPTH steps ahead of its use (global loads need that distance with
// one wave per SIMD), and forms the product on the way; only the terminal term of the last knot keeps a register.
enum : int { TILE_LM_QX = 1, TILE_LM_DN = 2, TILE_LM_REGEN = 4, TILE_LM_VP = 8, TILE_LM_ALL = 15, TILE_LM_VPG = 16, TILE_LM_QXR = 32 };
constexpr int TILE_QXR_DEPTH = 12;
constexpr int tile_lds_arrays(int lm) { return ((lm & TILE_LM_QX) ? 1 : 0) + ((lm & TILE_LM_DN) ? 1 : 0) + ((lm & TILE_LM_VP) ? 1 : 0); }
constexpr int tile_reg_arrays(int lm) { return 5 - tile_lds_arrays(lm) - ((lm & TILE_LM_VPG) ? 1 : 0) - ((lm & TILE_LM_QXR) ? 1 : 0); }
// (w = 0: HALF rows -- an instance with nx+nu <= 8 takes 8 lanes, two instances share a DPP row; see admm_tile_kernel)
constexpr int tile_lds_slot(int nx, int nu, int w) { return (w == 0 ? 8 : 4 / w) * (nx + nu) + 1; }
// bytes of wave-private LDS: bound tables (unless UB), the trajectory (unless REGEN), the offloaded arrays
constexpr long tile_lds_bytes(int nx, int nu, int n, int w, int r, int lm, bool ub) {
    return 8L * (2L * (ub ? 2 : n) * 16 * (w == 0 ? 1 : w) + ((lm & TILE_LM_REGEN) ? 1 : (n / r) * 64) +
                 tile_lds_arrays(lm) * (long)(n / r) * tile_lds_slot(nx, nu, w));
}
// two waves per SIMD when the L-long register arrays + the matrix rows fit 256 VGPRs AND eight waves' LDS fits the CU
constexpr int tile_waves_per_simd(int nx, int nu, int n, int r, int lm = 0, int w = 1) {
    return (2 * (tile_reg_arrays(lm) * (n / r) + 2 * (nx + nu)) + 44 <= 276 &&      // measured (round 2): (12,8,30) at 274 gains, (20,2,30) at 282 loses to spills
            (lm == 0 || 8 * tile_lds_bytes(nx, nu, n, w, r, lm, true) <= 158 * 1024)) ? 2 : 1;
}

// waves per SIMD of a cone variant (five L-long arrays in registers, the trajectory, the bound tables and the three slack planes in
// LDS): two when the arrays + matrix rows fit 256 VGPRs (same measured threshold as the box forms) and eight waves' LDS fits the CU
constexpr long tile_soc_lds_bytes(int nx, int nu, int n, int w, int r, int soc, bool ub) {
    const int cr = ((soc & 2) ? nx : 0) + ((soc & 1) ? nu : 0), csr = (cr + 1) | 1, ipw = 4 / (w * r);
    return 8L * (2L * (ub ? 2 : n) * 16 * w + (n / r) * 64 + (long)(ipw * n + 1) * 3 * csr);
}
template <int V> struct TileIntTag { static constexpr int value = V; };
template <bool C, class A, class B> struct TileSelect { typedef A type; };
template <class A, class B> struct TileSelect<false, A, B> { typedef B type; };
#ifndef TINYMPC_TILE_SOC_WAVES
#define TINYMPC_TILE_SOC_WAVES 0                   // (experiments: 1 / 2 instead of the rule)
#endif
constexpr int tile_soc_waves(int nx, int nu, int n, int w, int r, int soc, bool ub) {
    if (TINYMPC_TILE_SOC_WAVES > 0) return TINYMPC_TILE_SOC_WAVES;
    return (2 * (5 * (n / r) + 2 * (nx + nu)) + 44 <= 276 && 8 * tile_soc_lds_bytes(nx, nu, n, w, r, soc, ub) <= 158 * 1024) ? 2 : 1;
}

// SOC: second-order-cone slacks (admm.cpp:102-135, 228-235); bit 0: the input family's cone slack is on, bit 1: the state family's.
// This variant is never compiled in, it is instantiated at run time (jit.hpp) when a wide / long shape has a cone switched on.
// The slack lives in LDS planes and the cone step is TRANSPOSED, as in the one-row kernel (admm_kernel.hip.h): per instance and
// global slot three planes of one cell per row of the families that are on -- W (x + gc between the forward sweep and the cone
// step, then vcnew - gc: what the next backward sweep adds to the linear cost), GC (gc | yc), VC (vcnew | zcnew of the cells that
// belong to an item) -- and lane t of an instance's lanes takes (cone, knot) item t of a pass: ONE inside test / square root /
// division sequence per pass of LPI items instead of three shuffles and one per knot (round 4: the per-knot form ran at a
// quarter of the box form's iteration rate, profiles/r04_tile_variants_bench.md).
// LIN (bit 0 static, bit 1 time-varying half-spaces, admm.cpp:137-211) / KMAX as in admm_kernel.hip.h: two more L-long
// arrays per family; a'z is the lane-local product summed over the tile's rows by the same FMA chain against ones.
// UB: the box is the same at every knot -- a lane's two bounds (and the dummy slot's) live in registers, no LDS read per slot.
// W = 1 shapes compiled with TINYMPC_FUSED_NX / _NU (csrc/Makefile) run the sweeps on the one-row kernel's fused step blocks
// (fused_backward_step / fused_forward_step: the lane-local instructions of a step sit in front of its DPP chain, no s_nop)
// and take that kernel's placement of the forward constant (d <- fma(res, nim, cf)).
// EXT (run-time instantiated like SOC / LIN; round 5): bit 0 = the launch forms the one-row kernel offers a closed-loop caller -- a
// reference-trajectory window per MPC step (SolveArgs::traj: work->Xref = Xref_total.block(0, k, nx, N),
// examples/quadrotor_tracking.cpp:85-88), reset_duals (work->y = 0, work->g = 0 before every solve, :92-93), cold starts that
// do not read the warm-start records and the store masks of one_shot; bit 1 = per-instance problem data (HET: the matrix rows
// and rho of the instance's own cache, riccati_kernel.hip.h, in the tile table layout).  All arrays in registers (LM = 0).
// (EXT forms hold all five arrays in registers AND re-load the matrix rows per instance: two waves per SIMD only with room to spare --
// (20,8,10) at the box forms' threshold spilt 320 B per lane and ran at half the shared-family form's rate)
constexpr int tile_ext_waves(int nx, int nu, int n, int r, int lm = 0, int w = 1) {
    return lm != 0 ? tile_waves_per_simd(nx, nu, n, r, lm, w) : (2 * (5 * (n / r) + 2 * (nx + nu)) + 44 <= 240 ? 2 : 1);
}
template <int NX, int NU, int N, int W, int R, int SOC = 0, int LIN = 0, int KMAX = LIN_KMAX, bool UB = false, int LM = 0, bool DYN = false, int EXT = 0>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(LIN ? 1 : (SOC ? tile_soc_waves(NX, NU, N, W == 0 ? 1 : W, R, SOC, UB) : (EXT ? tile_ext_waves(NX, NU, N, R, LM, W) : tile_waves_per_simd(NX, NU, N, R, LM, W))),
                                   LIN ? 1 : (SOC ? tile_soc_waves(NX, NU, N, W == 0 ? 1 : W, R, SOC, UB) : (EXT ? tile_ext_waves(NX, NU, N, R, LM, W) : tile_waves_per_simd(NX, NU, N, R, LM, W))))))
void admm_tile_kernel(const SolveArgs P) {
    constexpr bool LS = (LIN & 1) != 0, LT = (LIN & 2) != 0;
    constexpr bool HR = W == 0;                                        // half rows: two instances per DPP row (nx+nu <= 8)
    constexpr bool EXTF = (EXT & 1) != 0, HETX = (EXT & 2) != 0;
    // bit 0 forms: all arrays in registers, static tiles.  Per-instance data ALONE (EXT == 2) rides on any box form of the shape -- its
    // LDS-offload set, the trajectory regenerated, v|z in its record, dynamic slots: the instance's matrix rows are just other values
    // in the same registers -- so that a heterogeneous batch runs the form the sweep measured fastest for the shape
    static_assert(EXT == 0 || W >= 1, "EXT forms: whole DPP rows");
    // Bit 0 too, as long as QX is writable (not re-read from the reference record): the window goes into the QX array wherever it lives.
    // (one_shot on a form that streams v|z: the host hands the stream a scratch array, SolveArgs::vz_stream -- the record stays untouched.)
    // (A reference WINDOW needs a QX array of its own: the host never pairs a trajectory with a form that re-reads QX from the reference
    // record -- batch_dispatch.hip launch_tile --; reset_duals and one_shot launches without a window take those forms too, round 6.)
    constexpr int WW = HR ? 1 : W;                                     // 16-lane rows across the knot vector (table layout)
    constexpr int ROWL = HR ? 8 : 16 * WW;                             // lanes between the horizon rows of one instance
    constexpr int NZ = NX + NU, LW = 16 * WW, L = N / R, RPI = WW * R, LPI = RPI * (HR ? 8 : 16), IPW = 64 / LPI;
    constexpr bool TFUSED = W == 1 && LIN == 0 && fused_shape(NX, NU);
    constexpr int NB = UB ? 2 : N;                                     // UB: slots 0 and 1 speak for all
    constexpr bool QL = (LM & TILE_LM_QX) != 0, DL = (LM & TILE_LM_DN) != 0, VL_ = (LM & TILE_LM_VP) != 0, VG = (LM & TILE_LM_VPG) != 0,
                   QR = (LM & TILE_LM_QXR) != 0;
    static_assert(!(VL_ && VG) && !(QL && QR) && (!QR || VG), "v|z: LDS or its record, not both; QX likewise (QXR rides on the VPG pointers)");
    constexpr bool KEEPX = !(LM & TILE_LM_REGEN) || SOC || LIN != 0;    // the cone / half-space slacks of the next solve start from x|u
    constexpr int SLOT = tile_lds_slot(NX, NU, W);
    // DEFER (horizon split over R > 1 rows, trajectory kept in LDS): a sweep phase runs on ONE of the R horizon rows while the others
    // idle, so everything in it costs R times its instructions.  The slot update (slack, dual, residual maxima: 9-10 lane-local
    // instructions per slot) needs nothing from the sweep but x_i, which the sweep leaves in sX anyway -- it moves behind the
    // sweep, where ALL rows run it on their own slots at once.  Same operations on the same values: bit-identical.
    // Measured (profiles/r03_tile_forms_defer.md): +7-13 % on the wide R = 2 forms and on (8,8,50); neutral to -5 % on narrow shapes, whose
    // fused step blocks had the slot update's instructions fill DPP wait states for free -- so only from 16 rows on.
    constexpr bool DEFER = R > 1 && KEEPX && !SOC && LIN == 0 && NZ >= 16;
    static_assert(N % R == 0 && NZ <= (HR ? 8 : LW) && RPI <= 4 && (RPI == 1 || RPI == 2 || RPI == 4) && L >= 2, "tile shape");
    static_assert(!HR || (!SOC && LIN == 0), "half rows: box constraints only");
    using T = TileTab<WW>;
    const int lane = threadIdx.x & 63, row = lane >> (HR ? 3 : 4), j16 = lane & (HR ? 7 : 15);
    const int inst = row / RPI, sub = row % RPI, wrow = sub % WW, hrow = sub / WW;
    const int jj = wrow * 16 + j16;
    const bool is_state = jj < NX, is_input = jj >= NX && jj < NZ;
    const int li = jj < NZ ? (row / WW) * NZ + jj : (SLOT - 1);       // this lane's entry of a compact LDS slot (the dummy beyond nx+nu)

    __shared__ double sLo[NB * LW];
    __shared__ double sHi[NB * LW];
    __shared__ double sX[KEEPX ? (N / R) * 64 : 1];    // x|u trajectory of this wave: only a rolling value in the sweep, kept for the output
    __shared__ double sQ[QL ? L * SLOT : 1];           // LM bit 0: QX
    __shared__ double sD[DL ? L * SLOT : 1];           // LM bit 1: Dn
    __shared__ double sV[VL_ ? L * SLOT : 1];          // LM bit 3: v|z (see the forward sweep)
    for (int e = lane; e < NB * LW; e += 64) {
        sLo[e] = P.tab[T::BOUNDS + e];
        sHi[e] = P.tab[T::BOUNDS + N * LW + e];
    }
    __shared__ double sLin[LS ? 3 * KMAX * LW : 1];
    __shared__ double sTLin[LT ? 3 * N * KMAX * LW : 1];
    if constexpr (LS) for (int e = lane; e < 3 * KMAX * LW; e += 64) sLin[e] = P.tab[T::lin_offset(N) + e];
    if constexpr (LT) for (int e = lane; e < 3 * N * KMAX * LW; e += 64) sTLin[e] = P.tab[T::tlin_offset(N, KMAX) + e];
    double mb[NZ], mf1[NX], mf2[NU];
#pragma unroll
    for (int k = 0; k < NZ; ++k) mb[k] = P.tab[T::MB + k * LW + jj];
#pragma unroll
    for (int k = 0; k < NX; ++k) mf1[k] = P.tab[T::MF1 + k * LW + jj];
#pragma unroll
    for (int k = 0; k < NU; ++k) mf2[k] = P.tab[T::MF2 + (NX + k) * LW + jj];
    typedef typename TileSelect<HETX, double, const double>::type FamilyScalar;   // (HETX: replaced by the instance's own at every load)
    FamilyScalar cb = P.tab[T::VEC + VEC_CB * LW + jj];
    FamilyScalar cf = P.tab[T::VEC + VEC_CF * LW + jj];
    FamilyScalar qr = P.tab[T::VEC + VEC_QR * LW + jj];
    const double smask = P.tab[T::VEC + VEC_SMASK * LW + jj];
    const double nim = P.tab[T::VEC + VEC_NIM * LW + jj];
    double socmask = 0.0;
    int cone_base = -1;
    auto g0_of = [](const int h) { return h * L; };
    if constexpr (SOC) {
        socmask = P.tab[T::VEC + VEC_SOCFLAG * LW + jj];               // 1.0 on the rows of a family whose cone slack is on
        cone_base = (int)P.tab[T::VEC + VEC_CONE_BASE * LW + jj];      // first ROW of this row's cone, or -1
    }
    bool lin_lane = false, tlin_lane = false;
    if constexpr (LS) lin_lane = P.tab[T::VEC + VEC_LINFLAG * LW + jj] != 0.0;
    if constexpr (LT) tlin_lane = P.tab[T::VEC + VEC_TLINFLAG * LW + jj] != 0.0;
    // (which families: uniform over the wave)
    bool lin_x_on = false, lin_u_on = false, tlin_x_on = false, tlin_u_on = false;
    if constexpr (LS) { lin_x_on = P.tab[T::VEC + VEC_LINFLAG * LW] != 0.0; lin_u_on = P.tab[T::VEC + VEC_LINFLAG * LW + NX] != 0.0; }
    if constexpr (LT) { tlin_x_on = P.tab[T::VEC + VEC_TLINFLAG * LW] != 0.0; tlin_u_on = P.tab[T::VEC + VEC_TLINFLAG * LW + NX] != 0.0; }
    // LIN: the half-space slacks live in LDS as well -- per set (static | time-varying) two planes of one cell per row and global
    // slot: V (x + gl between the forward sweep and the projection step, then vlnew) and G (gl); the backward sweep adds
    // -rho (V - G) (admm.cpp:272 ...).  The projections are TRANSPOSED like the cone step: one lane takes a whole (knot, family)
    // column, applies the family's half-spaces to it one after the other in its registers (a'z as the reference forms it: products
    // rounded, summed in row order; project_hyperplane only when violated, admm.cpp:148-173, 186-211) and writes vlnew and
    // gl = (x + gl) - vlnew back -- N + N - 1 columns per instance and set instead of two DPP chains per knot, half-space and slot.
    constexpr int CSL = LIN ? ((NZ + 1) | 1) : 1;                        // cells of a slot (+ the pad of the lanes beyond nx+nu), odd
    __shared__ double sLV[LS ? IPW * N * CSL : 1], sLG[LS ? IPW * N * CSL : 1];
    __shared__ double sTV[LT ? IPW * N * CSL : 1], sTG[LT ? IPW * N * CSL : 1];
    const int cl0 = (inst * N + hrow * L) * CSL + (jj < NZ ? jj : NZ);   // this lane's cell of its first own slot (slot l: + l * CSL)
    // one pass of columns: family rows [R0, R0 + NF), columns (= global slots) S0 + t of this lane's instance, t over the passes
    auto project_columns = [&](auto nf_tag, const int R0, const int S0, double* pv, double* pg, const double* tab, const int tab_stride, const int nk) {
#pragma clang fp contract(off)
        constexpr int NF = decltype(nf_tag)::value;
        const int t = lane - inst * LPI;
        for (int c = S0 + t; c < N; c += LPI) {
            const int at = (inst * N + c) * CSL + R0;
            const double* tk = tab + c * tab_stride;
            double z[NF], t0[NF];
#pragma unroll
            for (int r = 0; r < NF; ++r) { z[r] = pv[at + r]; t0[r] = z[r]; }
            for (int k = 0; k < nk; ++k) {
                double cv = 0.0;
#pragma unroll
                for (int r = 0; r < NF; ++r) { const double pr = tk[k * LW + R0 + r] * z[r]; cv = cv + pr; }
                const double bk = tk[KMAX * LW + k * LW + R0];
                if (cv > bk) {
                    const double dist = (cv - bk) / tk[2 * KMAX * LW + k * LW + R0];
#pragma unroll
                    for (int r = 0; r < NF; ++r) { const double pr = dist * tk[k * LW + R0 + r]; z[r] = z[r] - pr; }
                }
            }
#pragma unroll
            for (int r = 0; r < NF; ++r) { pv[at + r] = z[r]; pg[at + r] = t0[r] - z[r]; }
        }
    };
    using NXTag = TileIntTag<NX>;
    using NUTag = TileIntTag<NU>;
    const bool soc_lane = socmask != 0.0, proj_lane = soc_lane && cone_base >= 0;
    // ---- cone slack planes (see the header): CR cells per slot and plane, one per row of the families that are on; every other
    // lane shares the pad cell CR (it only ever holds zeros); a slot is 3 CSR doubles, an ODD number: the item gathers of a pass
    // (stride = one slot) fall into distinct LDS banks.  + a dummy slot whose item (0, 0, 1) the lanes without an item project
    // (onto itself): no EXEC-mask region around the gather / scatter of a pass.
    constexpr int CR = ((SOC & 2) ? NX : 0) + ((SOC & 1) ? NU : 0);
    constexpr int CSR = SOC ? ((CR + 1) | 1) : 1, SLOT_C = 3 * CSR, PL_GC = CSR, PL_VC = 2 * CSR;
    constexpr int C_ROW0 = (SOC == 1) ? NX : 0;                         // the row that owns cell 0
    __shared__ double sC[SOC ? (IPW * N + 1) * SLOT_C : 1];
    const int cell = (soc_lane && jj >= C_ROW0 && jj - C_ROW0 < CR) ? jj - C_ROW0 : CR;
    const int cw0 = (inst * N + g0_of(hrow)) * SLOT_C + cell;           // W cell of this lane's first own slot (slot l: + l * SLOT_C)
    // (cone, knot) items of an instance, dealt out to its LPI lanes 16 ... 64 per pass: item t = cone by cone (ascending first
    // row), a state cone has N items (slots 0 .. N-1), an input cone N-1 (slots 1 .. N-1).  All instances of a wave share the layout.
    constexpr int SOC_ITEMS = ((SOC & 2) ? (NX / 3) * N : 0) + ((SOC & 1) ? (NU / 3) * (N - 1) : 0);
    constexpr int SOC_PASSES = SOC ? (SOC_ITEMS > 0 ? (SOC_ITEMS + LPI - 1) / LPI : 1) : 1;
    int item_at[SOC_PASSES];
    float item_mu[SOC_PASSES], item_rmu[SOC_PASSES];
    bool mu_pow2 = true;
    int soc_passes = 0;
    if constexpr (SOC) {
        if (lane < 3) {
            sC[IPW * N * SLOT_C + lane] = lane == 2 ? 1.0 : 0.0;
            sC[IPW * N * SLOT_C + PL_GC + lane] = 0.0;
            sC[IPW * N * SLOT_C + PL_VC + lane] = lane == 2 ? 1.0 : 0.0;
        }
        // first rows of the cones: instance 0's first horizon row holds row jj in lane jj
        const unsigned heads = (unsigned)(__builtin_amdgcn_ballot_w64(proj_lane && jj == cone_base && inst == 0 && hrow == 0) & 0xFFFFFFFFull);
        int total = 0;
        for (unsigned m = heads; m; m &= m - 1) total += (__builtin_ctz(m) < NX) ? N : N - 1;
#pragma unroll
        for (int p = 0; p < SOC_PASSES; ++p) {
            int t = p * LPI + (lane - inst * LPI);
            item_at[p] = IPW * N * SLOT_C; item_mu[p] = 1.0f;
            if (p * LPI < total) soc_passes = p + 1;
            for (unsigned m = heads; m; m &= m - 1) {
                const int hb = __builtin_ctz(m);
                const int cnt = hb < NX ? N : N - 1;
                if (t >= 0 && t < cnt) {
                    item_at[p] = (inst * N + t + (hb < NX ? 0 : 1)) * SLOT_C + (hb - C_ROW0);
                    item_mu[p] = (float)P.tab[T::VEC + VEC_CONE_MU * LW + hb];
                    t = -1;
                } else if (t >= 0) t -= cnt;
            }
            item_rmu[p] = 1.0f / item_mu[p];
            const unsigned mb_ = __float_as_uint(item_mu[p]);
            const bool p2 = (mb_ & 0x807FFFFFu) == 0u && (mb_ >> 23) >= 127u - 60u && (mb_ >> 23) <= 127u + 60u;
            if (__builtin_amdgcn_ballot_w64(!p2) != 0ull) mu_pow2 = false;
        }
    }
    FamilyScalar rho = P.rho;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double lo_u = sLo[LW + jj], hi_u = sHi[LW + jj], lo_u0 = sLo[jj], hi_u0 = sHi[jj];     // UB (N >= 2 always)

    const unsigned long long inst_mask = (LPI == 64) ? ~0ull : ((((1ull << (LPI & 63)) - 1ull)) << (inst * LPI));
    const int nsteps = P.steps > 1 ? P.steps : 1;
    const int g0 = hrow * L;                                           // first global slot of this row
    // (round 6) a launch over an index list -- the open instances a capped one-row launch left behind (SolveArgs::index / count /
    // iter_base: the split solve's tail, batch_dispatch.hip enqueue_split_solve): slot s of the launch is instance index[s], its solve
    // carries on at iteration iter_base from the state the capped launch stored.  For a box variant that IS a warm start whose
    // iteration counter does not begin at zero; the rows of the dynamic form refill one by one, so no row waits for its neighbours.
    // Only the forms that can be asked to do it carry the code (RSM): the dynamic one-row-layout forms (R = 1, trajectory regenerated, all
    // arrays in registers: the "<shape> 1 1 4" / "0 1 4" entries of tile_dims.txt) -- one more live register cost the 512-register box
    // forms of the long shapes scratch ((8,8,50): 36 -> 156 B per lane, 12.1 -> 13.0 ms) when every form had it.
    constexpr bool RSM = DYN && EXT == 0 && SOC == 0 && LIN == 0 && R == 1 && LM == TILE_LM_REGEN;
    const int ninst = (RSM && P.index) ? *P.count : P.batch;
    const int ntiles = (ninst + IPW - 1) / IPW;
    // ---- the state of this lane's instance (one instance per SLOT of RPI rows; IPW slots per wave)
    double G[L], VN[L], VP[(VL_ || VG) ? 1 : L], QX[(QL || QR) ? 1 : L], Dn[DL ? 1 : L];
    bool gcz[SOC_PASSES];                                              // SOC: the GC cells of this lane's item of pass p are known to be zero
    bool gc_own_dirty = false;                                         // SOC: this lane's own GC cells outside every item may hold a loaded value
    double ref_last = 0.0, x0v = 0.0, x1v = 0.0, x0_last = 0.0, rp = 0.0, rd = 0.0;   // x1v: slot 1 (x_1 | u_0) of the last sweep; x0_last: the x0 the last solve started from
    int b = 0, iter = 0, solved = 0, checked = 0, countdown = 0, step = 0;
    [[maybe_unused]] int iter0_ = 0;                                    // RSM: where this solve's counter began (SolveArgs::iter_base); else 0
    auto iter0 = [&]() -> int { if constexpr (RSM) return iter0_; else return 0; };
    double *vpp = nullptr, *vpp0 = nullptr;                              // VPG: see the load
    const double *rpp = nullptr, *rpp0 = nullptr;                        // QXR: the same column of the reference record
    double qx_term = 0.0;                                               // QXR: -(Xref[:,N-1]' Pinf) of this lane (admm.cpp:292)
    unsigned acc_iter = 0, acc_solved = 0;
    bool have = false;                                                  // this slot holds an instance that is not finished yet
    int next_tile = blockIdx.x;                                         // static assignment: tiles of IPW instances, grid stride
    bool exhausted = false;                                             // DYN: the work counter has run past the batch (wave-uniform)
    // One loop pass = one ADMM iteration of every slot that holds an instance.  A slot that has none takes the next instance
    // first (load) and hands its finished one back last (store); both are executed under the EXEC mask of that slot only.
    //   static (DYN = false): the wave moves on to its next tile when ALL its slots are done -- a wave runs as long as its
    //     slowest instance, the others idle (E[max of 4] / mean = 1.5 ... 2.5 on the config-5 cells);
    //   dynamic (DYN = true, persistent grid): a slot takes the next instance off ONE device-wide counter the moment it is free
    //     (one atomic per wave and pass that needs any) -- rows only idle in the tail of the launch.  Instances are independent, so
    //     who solves which changes nothing in the results.
    // terminal term -(Xref[:,N-1]' Pinf) on the last horizon row (admm.cpp:292); every lane of the instance takes part in the chain
    // (EXT forms only: the others keep their inline copy in the load and capture nothing here)
    auto terminal_term = [&]() {
        if constexpr (EXT != 0) {
            double pt[NX];
            const double* const tp = HETX ? P.het_tabs + (size_t)b * T::BOUNDS : P.tab;
#pragma unroll
            for (int k = 0; k < NX; ++k) pt[k] = tp[T::PT + k * LW + jj];
            const double xp = tile_matvec<W, 0, NX>(0.0, ref_last, pt);
            if constexpr (QR) qx_term = -xp;
            else if (hrow == R - 1 && is_state) { if constexpr (QL) sQ[(L - 1) * SLOT + li] = -xp; else QX[L - 1] = -xp; }
        }
    };
    // EXT bit 0, at the start of every solve (MPC step `step` of this launch): the reference window and the dual reset
    auto ext_begin_solve = [&]() {
        if constexpr (EXTF) {
            if (!QR && P.traj) {                                        // work->Xref = Xref_total.block(0, k, nx, N)
                const int k0 = P.traj_step0 + (P.traj_offsets ? P.traj_offsets[b] : 0) + step + g0;
                if (is_state) {
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        int kk = k0 + l;
                        kk = kk < P.traj_points ? kk : P.traj_points - 1;
                        const double r = P.traj[(size_t)kk * NX + jj];
                        if constexpr (QL) sQ[l * SLOT + li] = -(r * qr); else QX[l] = -(r * qr);      // admm.cpp:266 / :279
                        if (l == L - 1) ref_last = r;
                    }
                }
                terminal_term();
                if constexpr (QL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // (each lane reads back its own entries)
            }
            if (P.reset_duals) {                                        // work->y = 0; work->g = 0
#pragma unroll
                for (int l = 0; l < L; ++l) G[l] = 0.0;
            }
        }
    };
    for (;;) {
        bool fresh = false;
        if constexpr (DYN) {
            const bool need = !have && !exhausted;
            const unsigned long long m = __ballot(need && sub == 0 && j16 == 0);          // one leader lane per slot that needs work
            if (m != 0ull) {
                const int first = __ffsll((long long)m) - 1, n = __popcll(m);
                int base = 0;
                if (lane == first) base = atomicAdd(P.work_counter, n);
                base = __shfl(base, first);
                const int leader = (lane / LPI) * LPI;                     // leader lane of this lane's slot
                if (need) {
                    const int slot = base + __popcll(m & ((1ull << leader) - 1ull));
                    fresh = slot < ninst;
                    b = (RSM && fresh && P.index) ? P.index[slot] : slot;
                }
                if (base + n >= ninst) exhausted = true;
            }
        } else {
            if (__ballot(have) == 0ull && next_tile < ntiles) {                          // lock step: the whole wave moves on together
                const int slot = next_tile * IPW + inst;
                fresh = slot < ninst;
                b = slot;                                   // (index lists: dynamic forms only)
                next_tile += gridDim.x;
            }
        }
        if (fresh) {
            if constexpr (HETX) {                                  // this instance's own cache (A, B, Q, R, rho differ per instance)
                const double* het = P.het_tabs + (size_t)b * T::BOUNDS;
#pragma unroll
                for (int k = 0; k < NZ; ++k) mb[k] = het[T::MB + k * LW + jj];
#pragma unroll
                for (int k = 0; k < NX; ++k) mf1[k] = het[T::MF1 + k * LW + jj];
#pragma unroll
                for (int k = 0; k < NU; ++k) mf2[k] = het[T::MF2 + (NX + k) * LW + jj];
                cb = het[T::VEC + VEC_CB * LW + jj];
                cf = het[T::VEC + VEC_CF * LW + jj];
                qr = het[T::VEC + VEC_QR * LW + jj];
                rho = het[T::VEC + 8 * LW + jj];                   // VEC_RHO (riccati_kernel.hip.h)
            }
            // ---- load the instance record
            [[maybe_unused]] const bool cold = EXTF && P.cold;     // the warm-start records are known to be zero: not read
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int g = g0 + l;
                const bool valid = is_state || (is_input && g >= 1);
                const bool warm = valid && !cold;
                const size_t off = ((size_t)b * N + (is_state ? g : g - 1)) * NZ + jj;
                const double r = valid ? P.ref[off] : 0.0;
                VN[l] = warm ? P.slack[off] : 0.0;
                G[l] = warm ? P.dual[off] : 0.0;
                if constexpr (VL_) sV[l * SLOT + li] = warm ? P.slack_prev[off] : 0.0; else if constexpr (!VG) VP[l] = warm ? P.slack_prev[off] : 0.0;
                if constexpr (QL) sQ[l * SLOT + li] = -(r * qr); else if constexpr (!QR) QX[l] = -(r * qr);
                if constexpr (DL) sD[l * SLOT + li] = 0.0; else Dn[l] = 0.0;
                if constexpr (SOC) {
                    const double vc0 = (warm && soc_lane) ? P.prim[off] : 0.0;          // vcnew = x, zcnew = u (admm.cpp:352-357)
                    const double gc0 = (warm && soc_lane) ? P.cdual[off] : 0.0;
                    sC[cw0 + l * SLOT_C] = vc0 - gc0;
                    sC[cw0 + l * SLOT_C + PL_GC] = gc0;
                    sC[cw0 + l * SLOT_C + PL_VC] = vc0;
                }
                if constexpr (LS) { sLV[cl0 + l * CSL] = (warm && lin_lane) ? P.prim[off] : 0.0; sLG[cl0 + l * CSL] = (warm && lin_lane) ? P.ldual[off] : 0.0; }     // :361-365
                if constexpr (LT) { sTV[cl0 + l * CSL] = (warm && tlin_lane) ? P.prim[off] : 0.0; sTG[cl0 + l * CSL] = (warm && tlin_lane) ? P.tldual[off] : 0.0; }  // :370-374
                if (l == L - 1) ref_last = r;                          // only meaningful on the last horizon row
            }
            x0v = (hrow == 0 && is_state) ? P.x0[(size_t)b * NX + jj] : 0.0;
            if constexpr (VG) {
                // this lane's column of the instance's v|z record (slot l at + l NZ); lanes that hold no row, and the input lanes' dummy
                // slot 0, go to the pad behind the records (tiny_batch_setup: zeros, and zeros are all that is ever written there)
                double* const vzb = (EXTF && P.vz_stream) ? P.vz_stream : P.slack_prev;     // (a one-shot launch streams into a scratch array)
                double* const pad = vzb + (size_t)P.batch * N * NZ + (lane & 15);
                vpp = jj < NZ ? vzb + ((size_t)b * N + g0 - (is_input ? 1 : 0)) * NZ + jj : pad;
                vpp0 = (hrow == 0 && is_input) ? pad : vpp;
                if constexpr (QR) { rpp = P.ref + (vpp - vzb); rpp0 = P.ref + (vpp0 - vzb); }
            }
            if constexpr (EXT == 0) {   // terminal term -(Xref[:,N-1]' Pinf) on the last horizon row (admm.cpp:292)
                double pt[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) pt[k] = P.tab[T::PT + k * LW + jj];
                const double xp = tile_matvec<W, 0, NX>(0.0, ref_last, pt);
                if constexpr (QR) qx_term = -xp;
                else if (hrow == R - 1 && is_state) { if constexpr (QL) sQ[(L - 1) * SLOT + li] = -xp; else QX[L - 1] = -xp; }
            } else if (!(EXTF && P.traj)) terminal_term();          // (a reference window: formed at the start of every solve, below)
            if constexpr (QL || DL || VL_) {                           // (each lane only ever reads its own entries back: no barrier needed,
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the fence keeps the compiler from moving LDS reads above these writes)
            }

            have = true; step = 0; acc_iter = 0; acc_solved = 0; checked = 0; rp = 0.0; rd = 0.0; x1v = 0.0;
            if constexpr (SOC) gc_own_dirty = true;
        }
        if (__ballot(have) == 0ull) break;
        bool start = fresh;                                             // a solve begins: this instance's first, or its next fused MPC step
        if (have) {
            if (start) {
                if constexpr (RSM) iter0_ = step == 0 ? P.iter_base : 0;      // (a multiple of check_termination: the countdown restarts in phase)
                iter = iter0(); solved = 0; countdown = P.check_termination;
                if (iter0() > 0 && P.check_termination > 0) checked = 1;
                x0_last = x0v;
                ext_begin_solve();
                if constexpr (SOC) {
                    if (hrow == 0 && is_state && soc_lane) {               // x[:,0] = x0
                        sC[cw0] = x0v - sC[cw0 + PL_GC];
                        sC[cw0 + PL_VC] = x0v;
                    }
#pragma unroll
                    for (int p = 0; p < SOC_PASSES; ++p) gcz[p] = false;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                if constexpr (LS) { if (hrow == 0 && is_state && lin_lane) sLV[cl0] = x0v; }      // x[:,0] = x0 (admm.cpp:361-365)
                if constexpr (LT) { if (hrow == 0 && is_state && tlin_lane) sTV[cl0] = x0v; }     // (:370-374)
            }
            bool conv = false;
            if (iter < P.max_iter) {
                // ---- backward_pass_grad (admm.cpp:13-20): the horizon rows take turns, last row first
                double pcur = 0.0, qhi = 0.0;
#pragma unroll
                for (int ph = R - 1; ph >= 0; --ph) {
                    if (ph < R - 1) {                                  // p_{i+1} | r_i handed down from the row above
                        pcur = __shfl(pcur, (lane + ROWL) & 63);
                        qhi = __shfl(qhi, (lane + ROWL) & 63);
                    }
                    if (hrow == ph) {
                        // QX in LDS: read two steps ahead of its use (a lone wave per SIMD has nothing else to hide the LDS latency
                        // behind), and no further: the scheduling barrier at every step keeps the compiler from hoisting all L reads
                        // to the top of the sweep, which is what turns 3 register arrays into 1 KB of scratch per lane
                        double qa = 0.0, qb = 0.0;
                        if constexpr (QL) { qa = sQ[(L - 1) * SLOT + li]; qb = sQ[(L - 2) * SLOT + li]; }
                        // QXR: the reference record again, QD steps ahead (a ring of QD values in flight; the sweep is unrolled, so the
                        // ring index is a constant at every step)
                        constexpr int QD = L < TILE_QXR_DEPTH ? L : TILE_QXR_DEPTH;
                        double rq[QR ? QD : 1];
                        if constexpr (QR) {
#pragma unroll
                            for (int d = 0; d < QD; ++d) rq[d] = (L - 1 - d == 0 ? rpp0 : rpp)[(L - 1 - d) * NZ];
                        }
                        // SOC: vcnew - gc of slot l out of its W cell, two steps ahead of its use (one wave per SIMD: see QX above)
                        double wa = 0.0, wb = 0.0;
                        if constexpr (SOC) { wa = sC[cw0 + (L - 1) * SLOT_C]; wb = sC[cw0 + (L - 2) * SLOT_C]; }
                        // LIN: vlnew and gl of slot l out of their planes, likewise
                        double lva = 0.0, lvb = 0.0, lga = 0.0, lgb = 0.0, tva = 0.0, tvb = 0.0, tga = 0.0, tgb = 0.0;
                        if constexpr (LS) { lva = sLV[cl0 + (L - 1) * CSL]; lvb = sLV[cl0 + (L - 2) * CSL]; lga = sLG[cl0 + (L - 1) * CSL]; lgb = sLG[cl0 + (L - 2) * CSL]; }
                        if constexpr (LT) { tva = sTV[cl0 + (L - 1) * CSL]; tvb = sTV[cl0 + (L - 2) * CSL]; tga = sTG[cl0 + (L - 1) * CSL]; tgb = sTG[cl0 + (L - 2) * CSL]; }
#pragma unroll
                        for (int l = L - 1; l >= 0; --l) {
                            double wl = 0.0, lvl = 0.0, lgl = 0.0, tvl = 0.0, tgl = 0.0;
                            if constexpr (SOC) {
                                wl = wa; wa = wb;
                                if (l >= 2) wb = sC[cw0 + (l - 2) * SLOT_C];
                            }
                            if constexpr (LS) {
                                lvl = lva; lva = lvb; lgl = lga; lga = lgb;
                                if (l >= 2) { lvb = sLV[cl0 + (l - 2) * CSL]; lgb = sLG[cl0 + (l - 2) * CSL]; }
                            }
                            if constexpr (LT) {
                                tvl = tva; tva = tvb; tgl = tga; tga = tgb;
                                if (l >= 2) { tvb = sTV[cl0 + (l - 2) * CSL]; tgb = sTG[cl0 + (l - 2) * CSL]; }
                            }
                            if constexpr (SOC != 0 || LIN != 0) __builtin_amdgcn_sched_barrier(0);
                            double qxl;
                            if constexpr (QL) {
                                qxl = qa; qa = qb;
                                if (l >= 2) qb = sQ[(l - 2) * SLOT + li];
                            } else if constexpr (QR) {
                                const double r = rq[(L - 1 - l) % QD];
                                if (l - QD >= 0) rq[(L - 1 - l) % QD] = (l - QD == 0 ? rpp0 : rpp)[(l - QD) * NZ];
                                __builtin_amdgcn_sched_barrier(0);                      // (the load stays HERE: QD steps ahead of its use, no further)
                                qxl = -(r * qr);                                        // admm.cpp:266 / :279
                                if (l == L - 1) qxl = (hrow == R - 1 && is_state) ? qx_term : qxl;
                            } else qxl = QX[l];
                            double qlo = fma(-rho, VN[l] - G[l], qxl);                  // admm.cpp:267 | :280 | :293
                            if constexpr (SOC) qlo = fma(-rho, wl, qlo);                // :269 | :282 | :295
                            if constexpr (LS) qlo = fma(-rho, lvl - lgl, qlo);          // :272 | :285 | :298
                            if constexpr (LT) qlo = fma(-rho, tvl - tgl, qlo);          // :275 | :288 | :301
                            if (ph == R - 1 && l == L - 1) {
                                pcur = qlo;                                             // p_{N-1}
                            } else if constexpr (TFUSED) {
                                double q2, res;                                         // (q2 == qlo: the block forms it again in front of its chain)
                                if constexpr (SOC != 0) fused_backward_step_soc<NX, NU>(q2, res, VN[l], G[l], qxl, wl, rho, smask, cb, pcur, qhi, mb, mb + NX);
                                else fused_backward_step<NX, NU>(q2, res, VN[l], G[l], qxl, rho, smask, cb, pcur, qhi, mb, mb + NX);
                                pcur = res;
                                const double dn = fma(res, nim, cf);                    // input lanes: -d_i; state lanes: fdyn
                                if constexpr (DL) sD[l * SLOT + li] = dn; else Dn[l] = dn;
                                qlo = q2;
                            } else {
                                const double src = is_input ? qhi : pcur;
                                const double res = tile_matvec<W, 0, NZ>(fma(qlo, smask, cb), src, mb);
                                pcur = res;                                             // p_i | d_i
                                const double dn = HR ? fma(res, nim, cf) : res * nim;   // (half rows: the one-row kernel's placement of the forward constant, so that the two are interchangeable bit for bit)
                                if constexpr (DL) sD[l * SLOT + li] = dn; else Dn[l] = dn;
                            }
                            qhi = qlo;
                        }
                    }
                }
                if constexpr (VG) {
                    // (a cold start -- one_shot -- takes v|z as zero without reading it: vnew|znew were loaded as zeros, and only a
                    // launch's FIRST solve is cold: the later solves of a fused launch read what the stream left)
                    if (iter == iter0() && !(EXTF && P.cold && step == 0)) {   // a solve's first iteration: v|z of the solve before, from its record, into
                        __builtin_amdgcn_s_waitcnt(0);                 // the vnew|znew registers (dead behind the backward sweep)
#pragma unroll
                        for (int l = 0; l < L; ++l) VN[l] = (l == 0 ? vpp0 : vpp)[l * NZ];
                        // The pad behind the records is shared by EVERY instance's lanes without a row and by the input lanes' dummy slot 0:
                        // a diverged instance can leave non-finite values there.  The dummy slot of a real lane does not take them (one
                        // select per solve); the lanes without a row may -- nothing of theirs reaches a row lane (the chains broadcast lanes
                        // < nx+nu only) and they do not vote in the termination test (below).  (a select on all L slots here
                        // cost these 512-register forms 50-80 B/lane of scratch and (12,2,50) 23 %.)
                        if (hrow == 0 && is_input) VN[0] = 0.0;
                    }
                }
                // ---- forward_pass (admm.cpp:25-32) + slot updates, first row first
                double pmax = 0.0, dmax = 0.0, xcarry = 0.0;
#pragma unroll
                for (int ph = 0; ph < R; ++ph) {
                    if (ph > 0) xcarry = __shfl(xcarry, (lane - ROWL) & 63);    // x_g | u_{g-1} of this row's first slot
                    if (hrow == ph) {
                        double xcur = (ph > 0) ? xcarry : x0v;         // x_g | u_{g-1} of the slot being processed
                        // the box of slot l+1 is read from LDS while slot l is worked on (as admm_kernel.hip.h does): a lone wave
                        // per SIMD has nothing else to hide the LDS latency behind
                        double lo_c = UB ? (ph == 0 ? lo_u0 : lo_u) : sLo[(ph * L) * LW + jj], hi_c = UB ? (ph == 0 ? hi_u0 : hi_u) : sHi[(ph * L) * LW + jj];
                        double da = 0.0, db = 0.0, va = 0.0, vb = 0.0;   // Dn / v|z in LDS: read two steps ahead (see the backward sweep)
                        if constexpr (DL) { da = sD[li]; db = sD[SLOT + li]; }
                        if constexpr (VL_) { va = sV[li]; vb = sV[SLOT + li]; }
                        double ga = 0.0, gb = 0.0;                      // SOC: gc of slot l out of its GC cell, likewise
                        if constexpr (SOC) { ga = sC[cw0 + PL_GC]; gb = sC[cw0 + SLOT_C + PL_GC]; }
                        double fla = 0.0, flb = 0.0, fta = 0.0, ftb = 0.0;   // LIN: gl | gl_tv of slot l
                        if constexpr (LS) { fla = sLG[cl0]; flb = sLG[cl0 + CSL]; }
                        if constexpr (LT) { fta = sTG[cl0]; ftb = sTG[cl0 + CSL]; }
#pragma unroll
                        for (int l = 0; l < L; ++l) {
                            const int g = ph * L + l;
                            double lo_n = lo_u, hi_n = hi_u;
                            double dcur = 0.0;
                            if constexpr (DL) {
                                dcur = da; da = db;
                                if (l + 2 < L) db = sD[(l + 2) * SLOT + li];
                            }
                            double vcur = 0.0;
                            if constexpr (VL_) {
                                vcur = va; va = vb;
                                if (l + 2 < L) vb = sV[(l + 2) * SLOT + li];
                            }
                            if constexpr (!UB) {
                                lo_n = (l + 1 < L) ? sLo[(g + 1) * LW + jj] : 0.0; hi_n = (l + 1 < L) ? sHi[(g + 1) * LW + jj] : 0.0;
                            }
                            double gcl = 0.0;
                            if constexpr (SOC) {
                                gcl = ga; ga = gb;
                                if (l + 2 < L) gb = sC[cw0 + (l + 2) * SLOT_C + PL_GC];
                            }
                            double gll = 0.0, gtl = 0.0;
                            if constexpr (LS) { gll = fla; fla = flb; if (l + 2 < L) flb = sLG[cl0 + (l + 2) * CSL]; }
                            if constexpr (LT) { gtl = fta; fta = ftb; if (l + 2 < L) ftb = sTG[cl0 + (l + 2) * CSL]; }
                            if constexpr (!UB || DL || VL_ || SOC != 0 || LIN != 0) __builtin_amdgcn_sched_barrier(0);   // (pins the reads ABOVE this step: the compiler otherwise sinks them to their use)
                            const double xi = xcur;
                            if constexpr (KEEPX) sX[l * 64 + lane] = xi;
                            if constexpr (DEFER) {                                       // the chains only; the slot update follows the sweep
                                if (g < N - 1) {
                                    double dnl;
                                    if constexpr (DL) dnl = dcur; else dnl = Dn[l];
                                    double xn;
                                    if constexpr (TFUSED) {                              // (the one-row kernel's placement of the forward constant)
                                        const double t = tile_matvec<W, 0, NX>(dnl, xi, mf1);
                                        xn = tile_matvec<W, NX, NZ>(t, t, mf2);
                                    } else {
                                        const double t = tile_matvec<W, 0, NX>(dnl, xi, mf1);
                                        xn = tile_matvec<W, NX, NZ>(HR ? t : t + cf, t, mf2);
                                    }
                                    if (l + 1 < L) xcur = xn; else xcarry = xn;
                                    if (g == 0) x1v = xn;
                                }
                                lo_c = lo_n; hi_c = hi_n;
                                continue;
                            }
                            double tt, vn;
                            if constexpr (TFUSED) {
                                if (g < N - 1) {
                                    double xn, t;
                                    if constexpr (DL) t = dcur; else t = Dn[l];
                                    fused_forward_step<NX, NU>(tt, vn, t, xn, xi, G[l], lo_c, hi_c, mf1, mf2);
                                    if (l + 1 < L) xcur = xn; else xcarry = xn;
                                    if (g == 0) x1v = xn;
                                } else {
                                    tt = xi + G[l];
                                    vn = vmin64(hi_c, vmax64(lo_c, tt));
                                }
                            } else {
                                if (g < N - 1) {
                                    double dnl;
                                    if constexpr (DL) dnl = dcur; else dnl = Dn[l];
                                    const double t = tile_matvec<W, 0, NX>(dnl, xi, mf1);           // A x_i | u_i
                                    const double xn = tile_matvec<W, NX, NZ>(HR ? t : t + cf, t, mf2);   // + f + B u_i | u_i
                                    if (l + 1 < L) xcur = xn; else xcarry = xn;
                                    if (g == 0) x1v = xn;
                                }
                                tt = xi + G[l];
                                vn = vmin64(hi_c, vmax64(lo_c, tt));
                            }
                            lo_c = lo_n; hi_c = hi_n;
                            pmax = vmax_abs64(pmax, xi - vn);
                            if constexpr (VL_) dmax = vmax_abs64(dmax, vcur - vn);
                            else if constexpr (VG) {
                                dmax = vmax_abs64(dmax, VN[l] - vn);
                                (l == 0 ? vpp0 : vpp)[l * NZ] = VN[l];
                            } else dmax = vmax_abs64(dmax, VP[l] - vn);
                            G[l] = tt - vn;
                            VN[l] = vn;
                            // vcnew = x + gc on every row of a family whose cone slack is on (:102-109), projected after the sweep, one
                            // lane per (cone, knot): the cone step below.  A cell outside every item keeps this value as its vcnew.
                            // (only the lanes that own a cell write: the pad the others share must stay zero whatever an instance diverges to)
                            if constexpr (SOC) { if (soc_lane) sC[cw0 + l * SLOT_C] = fma(xi, socmask, gcl); }
                            // vlnew = x + gl on the rows of a family whose half-space slack is on (:139 / :144 / :177 / :182), projected after the
                            // sweep, one lane per (knot, family) column
                            if constexpr (LS) sLV[cl0 + l * CSL] = (lin_lane && (is_state || g >= 1)) ? (xi + gll) : 0.0;
                            if constexpr (LT) sTV[cl0 + l * CSL] = (tlin_lane && (is_state || g >= 1)) ? (xi + gtl) : 0.0;
                        }
                    }
                }
                if constexpr (DEFER) {
                    // ---- update_slack + update_dual + residual maxima (admm.cpp:81-98, 219-225, 314-317) of every row's own slots
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (sX: each lane reads back what it wrote itself)
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        const int g = g0 + l;
                        const double xi = sX[l * 64 + lane];
                        const double lo_c = UB ? (g == 0 ? lo_u0 : lo_u) : sLo[g * LW + jj], hi_c = UB ? (g == 0 ? hi_u0 : hi_u) : sHi[g * LW + jj];
                        const double tt = xi + G[l];
                        const double vn = vmin64(hi_c, vmax64(lo_c, tt));
                        pmax = vmax_abs64(pmax, xi - vn);
                        if constexpr (VL_) dmax = vmax_abs64(dmax, sV[l * SLOT + li] - vn);
                        else if constexpr (VG) {
                            dmax = vmax_abs64(dmax, VN[l] - vn);
                            (l == 0 ? vpp0 : vpp)[l * NZ] = VN[l];
                        } else dmax = vmax_abs64(dmax, VP[l] - vn);
                        G[l] = tt - vn;
                        VN[l] = vn;
                    }
                }
                if constexpr (LIN != 0) {
                    // ---- half-space projections (admm.cpp:137-211) + their dual update (:239-254), transposed (see project_columns)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if constexpr (LS) {
                        if (lin_x_on) project_columns(NXTag{}, 0, 0, sLV, sLG, sLin, 0, P.n_lin);
                        if (lin_u_on) project_columns(NUTag{}, NX, 1, sLV, sLG, sLin, 0, P.n_lin);
                    }
                    if constexpr (LT) {
                        if (tlin_x_on) project_columns(NXTag{}, 0, 0, sTV, sTG, sTLin, 3 * KMAX * LW, P.n_tlin);
                        if (tlin_u_on) project_columns(NUTag{}, NX, 1, sTV, sTG, sTLin, 3 * KMAX * LW, P.n_tlin);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if constexpr (SOC) {
                    // ---- cone step (admm.cpp:112-135, 228-235), transposed: x + gc of every slot went to the W plane above; lane t of
                    // pass p gathers the three components of its (cone, knot) item, projects them and writes the item's cells
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int p = 0; p < SOC_PASSES; ++p) {
                        if (p > 0 && p >= soc_passes) break;               // wave-uniform
                        const int at = item_at[p];
                        const double s0 = sC[at], s1 = sC[at + 1], s2 = sC[at + 2];
                        if (soc_all_inside(s0, s1, s2, item_mu[p])) {
                            // no cone of the wave's iterating instances is active: vcnew = x + gc bit for bit, gc = 0, and vcnew - gc is
                            // what the forward sweep left in W.  VC takes its copy; the GC cells are zeroed once per solve
                            sC[at + PL_VC] = s0; sC[at + PL_VC + 1] = s1; sC[at + PL_VC + 2] = s2;
                            if (!gcz[p]) {
                                sC[at + PL_GC] = 0.0; sC[at + PL_GC + 1] = 0.0; sC[at + PL_GC + 2] = 0.0;
                                gcz[p] = true;
                            }
                        } else {
                            double r0, r1, r2;
                            soc_project3(s0, s1, s2, item_mu[p], item_rmu[p], mu_pow2, r0, r1, r2);
                            const double g0c = s0 - r0, g1c = s1 - r1, g2c = s2 - r2;   // :229 / :234  (gc + x) - vcnew
                            sC[at + PL_VC] = r0; sC[at + PL_VC + 1] = r1; sC[at + PL_VC + 2] = r2;
                            sC[at + PL_GC] = g0c; sC[at + PL_GC + 1] = g1c; sC[at + PL_GC + 2] = g2c;
                            sC[at] = r0 - g0c; sC[at + 1] = r1 - g1c; sC[at + 2] = r2 - g2c;   // vcnew - gc: the next backward sweep's term
                            gcz[p] = false;
                        }
                    }
                    // this lane's own cells outside every item (a family row that belongs to no cone): vcnew = x + gc, so
                    // gc = (x + gc) - vcnew = 0 from the first iteration on, whatever the warm start held there
                    if (gc_own_dirty) {
                        if (soc_lane && !proj_lane) {
#pragma unroll
                            for (int l = 0; l < L; ++l) sC[cw0 + l * SLOT_C + PL_GC] = 0.0;
                        }
                        gc_own_dirty = false;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                iter += 1;
                if (countdown > 0 && --countdown == 0) {
                    countdown = P.check_termination;
                    checked = 1;
                    rp = pmax;
                    rd = dmax * rho;
                    const bool ok = ((rp < P.tol_pri) && (rd < P.tol_dua)) || jj >= NZ;        // (lanes without a row: their residuals are all zeros, or pad leftovers)
                    const unsigned long long bal = __ballot(ok);
                    conv = (bal & inst_mask) == inst_mask;
                }
                if (!conv) {
#pragma unroll
                    for (int l = 0; l < L; ++l) { if constexpr (VL_) sV[l * SLOT + li] = VN[l]; else if constexpr (!VG) VP[l] = VN[l]; }      // :445-446 (L LDS stores, no VALU work)
                }
            }
            if (conv || iter >= P.max_iter) {                            // this solve is over (admm.cpp:431-441 | :448-454)
                solved = conv ? 1 : 0;
                if constexpr (VG) {
                    if (!conv && iter > iter0()) {                       // out of iterations: v = vnew was the last thing that happened (:445-446)
#pragma unroll
                        for (int l = 0; l < L; ++l) (l == 0 ? vpp0 : vpp)[l * NZ] = VN[l];
                    }
                }
                acc_iter += (unsigned)(iter - iter0());
                acc_solved += (unsigned)solved;
                if (nsteps > 1) {
                    if (P.iter_log && sub == 0 && j16 == 0) P.iter_log[(size_t)step * P.batch + b] = solved ? iter : -iter;
                    if (P.u0_log && hrow == 0 && is_input && iter > 0) P.u0_log[((size_t)step * P.batch + b) * NU + (jj - NX)] = x1v;
                    // plant step x0 <- A x0 + B u_0 + f = the forward pass' x_1 (slot 1 of the first horizon row; L >= 2)
                    if (iter > 0 && step + 1 < nsteps) x0v = (hrow == 0 && is_state) ? x1v : 0.0;
                }
                step += 1;
                if (step < nsteps) {
                    // the next fused MPC step of the same instance starts in the next pass
                    if constexpr (RSM) iter0_ = 0;
                    iter = 0; solved = 0; countdown = P.check_termination;
                    x0_last = x0v;
                    ext_begin_solve();
                    if constexpr (SOC) {
                        if (soc_lane) {                                    // vcnew = x, zcnew = u of the previous solve (admm.cpp:352-357); x[:,0] = x0
#pragma unroll
                            for (int l = 0; l < L; ++l) {
                                const double vc0 = (l == 0 && hrow == 0 && is_state) ? x0v : sX[l * 64 + lane];
                                sC[cw0 + l * SLOT_C] = vc0 - sC[cw0 + l * SLOT_C + PL_GC];
                                sC[cw0 + l * SLOT_C + PL_VC] = vc0;
                            }
                        }
#pragma unroll
                        for (int p = 0; p < SOC_PASSES; ++p) gcz[p] = false;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    }
                    if constexpr (LS) {                                        // vlnew = x, zlnew = u (admm.cpp:361-365); x[:,0] = x0
#pragma unroll
                        for (int l = 0; l < L; ++l) sLV[cl0 + l * CSL] = lin_lane ? ((l == 0 && hrow == 0 && is_state) ? x0v : sX[l * 64 + lane]) : 0.0;
                    }
                    if constexpr (LT) {                                        // vlnew_tv = x, zlnew_tv = u (admm.cpp:370-374)
#pragma unroll
                        for (int l = 0; l < L; ++l) sTV[cl0 + l * CSL] = tlin_lane ? ((l == 0 && hrow == 0 && is_state) ? x0v : sX[l * 64 + lane]) : 0.0;
                    }
                    if constexpr (LIN != 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                } else {
                if constexpr (!KEEPX) {
                    // ---- the x|u trajectory, regenerated: forward_pass (admm.cpp:25-32) once more with the d of the last iteration and
                    // the x0 that solve started from -- the same instructions on the same inputs as the last sweep, bit for bit
                    if (iter > 0) {
                        double xcarry = 0.0;
#pragma unroll
                        for (int ph = 0; ph < R; ++ph) {
                            if (ph > 0) xcarry = __shfl(xcarry, (lane - ROWL) & 63);
                            if (hrow == ph) {
                                double xcur = (ph > 0) ? xcarry : x0_last;
#pragma unroll
                                for (int l = 0; l < L; ++l) {
                                    const int g = ph * L + l;
                                    const bool valid = is_state || (is_input && g >= 1);
                                    const size_t off = ((size_t)b * N + (is_state ? g : g - 1)) * NZ + jj;
                                    const double xi = xcur;
                                    if (valid && (!EXTF || (P.store_mask & 1) || ((P.store_mask & 32) && g <= 1))) P.prim[off] = xi;
                                    if (g < N - 1) {
                                        double xn, dnl;
                                        if constexpr (DL) dnl = sD[l * SLOT + li]; else dnl = Dn[l];
                                        if constexpr (TFUSED) {
                                            double tt, vn, t = dnl;
                                            fused_forward_step<NX, NU>(tt, vn, t, xn, xi, 0.0, 0.0, 0.0, mf1, mf2);
                                        } else {
                                            const double t = tile_matvec<W, 0, NX>(dnl, xi, mf1);
                                            xn = tile_matvec<W, NX, NZ>(HR ? t : t + cf, t, mf2);
                                        }
                                        if (l + 1 < L) xcur = xn; else xcarry = xn;
                                    }
                                }
                            }
                        }
                    } else if (hrow == 0 && is_state) {
                        P.prim[((size_t)b * N) * NZ + jj] = x0v;           // max_iter = 0: only x[:,0] = x0 is set
                    }
                }
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const int g = g0 + l;
                    const bool valid = is_state || (is_input && g >= 1);
                    const size_t off = ((size_t)b * N + (is_state ? g : g - 1)) * NZ + jj;
                    if (valid) {
                        // max_iter = 0: the sweeps never ran, x[:,1:] and u keep what they held (only x[:,0] = x0 is set)
                        // EXT: SolveArgs::store_mask as in the one-row kernel (bit 0 x|u, 5 its first knot only, 1 vnew|znew, 2 g|y, 3 v|z, 4 the slacks)
                        const int sm = EXTF ? P.store_mask : 63;
                        if constexpr (KEEPX) {
                        if ((sm & 1) || ((sm & 32) && g <= 1)) {
                        if (iter > 0) P.prim[off] = sX[l * 64 + lane];
                        else if (g == 0 && is_state) P.prim[off] = x0v;
                        }
                        }
                        if (sm & 2) P.slack[off] = VN[l];
                        if (sm & 4) P.dual[off] = G[l];
                        if (sm & 8) { if constexpr (VL_) P.slack_prev[off] = sV[l * SLOT + li]; else if constexpr (!VG) P.slack_prev[off] = VP[l]; }
                        if constexpr (SOC) {
                            // vcnew: the VC plane where the cell belongs to an item (or no iteration ran: what the solve started from),
                            // else what the last forward sweep left in the W plane
                            if (soc_lane && (sm & 16)) {
                                P.cslack[off] = sC[cw0 + l * SLOT_C + ((proj_lane || iter == 0) ? PL_VC : 0)];
                                P.cdual[off] = sC[cw0 + l * SLOT_C + PL_GC];
                            }
                        }
                        if constexpr (LS) { if (lin_lane && (sm & 16)) { P.lslack[off] = sLV[cl0 + l * CSL]; P.ldual[off] = sLG[cl0 + l * CSL]; } }
                        if constexpr (LT) { if (tlin_lane && (sm & 16)) { P.tlslack[off] = sTV[cl0 + l * CSL]; P.tldual[off] = sTG[cl0 + l * CSL]; } }
                    }
                }
                if (P.x0_next && iter > 0 && hrow == 0 && is_state) P.x0_next[(size_t)b * NX + jj] = x1v;
                // residual maxima over the instance's lanes (state rows / input rows separately)
                double ps = is_state ? rp : 0.0, pi = is_input ? rp : 0.0, ds = is_state ? rd : 0.0, di = is_input ? rd : 0.0;
#pragma unroll
                for (int off = LPI / 2; off >= 1; off >>= 1) {
                    ps = fmax(ps, __shfl_xor(ps, off)); pi = fmax(pi, __shfl_xor(pi, off));
                    ds = fmax(ds, __shfl_xor(ds, off)); di = fmax(di, __shfl_xor(di, off));
                }
                if (sub == 0 && j16 == 0) {
                    P.status[b] = make_int4(iter, solved, solved ? 1 : 11, checked);
                    *reinterpret_cast<double4*>(P.resid + (size_t)b * 4) = make_double4(ps, pi, ds, di);
                    if (P.accum) {
                        uint2 ac = P.accum[b];
                        ac.x += acc_iter;
                        ac.y += acc_solved;
                        P.accum[b] = ac;
                    }
                }
                    have = false;
                }
            }
        }
    }
}

// The compiled-in form of a tile shape (csrc/Makefile): as many arrays leave the register file as one wave's LDS share allows.
// Preference by accesses per slot and iteration saved: v|z (read + write + the copy), Dn (write + read), QX (read).  One wave per
// SIMD must keep four waves per CU resident (160 KB / 4, a little less for the allocation granule); two waves per SIMD eight.
// No kernel at all (nullptr) when not even the trajectory-free form fits the static limit of 64 KB -- a (W, R) entry further down
// tile_dims.txt then serves.
constexpr long TILE_LDS_STATIC_LIMIT = 64 * 1024 - 512;
constexpr int tile_best_lm(int nx, int nu, int n, int w, int r, bool ub) {
    const int cand[4] = {TILE_LM_REGEN | TILE_LM_VP | TILE_LM_DN | TILE_LM_QX, TILE_LM_REGEN | TILE_LM_VP | TILE_LM_DN, TILE_LM_REGEN | TILE_LM_VP,
                         TILE_LM_REGEN};
    for (int i = 0; i < 4; ++i) {
        const long bytes = tile_lds_bytes(nx, nu, n, w, r, cand[i], ub);
        const int waves = tile_waves_per_simd(nx, nu, n, r, cand[i], w);
        if (bytes <= TILE_LDS_STATIC_LIMIT && bytes * 4 * waves <= 158 * 1024) return cand[i];
    }
    return -1;
}
typedef void (*TileKernelFn)(const SolveArgs);
template <int NX, int NU, int N, int W, int R, int LMREQ, bool UB, bool DYN = false>
constexpr TileKernelFn tile_kernel_or_null() {
    // LMREQ (6th column of tile_dims.txt): 99 = the rule above; else that very set, if one wave's static LDS holds it
    constexpr int lm = LMREQ == 99 ? tile_best_lm(NX, NU, N, W, R, UB) : (tile_lds_bytes(NX, NU, N, W, R, LMREQ, UB) <= TILE_LDS_STATIC_LIMIT ? LMREQ : -1);
    if constexpr (lm >= 0) return admm_tile_kernel<NX, NU, N, W, R, 0, 0, LIN_KMAX, UB, (lm >= 0 ? lm : 0), DYN>;
    else return nullptr;
}

}  // namespace tinympc_amd
