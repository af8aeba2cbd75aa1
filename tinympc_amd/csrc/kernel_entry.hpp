// kernel_entry.hpp -- registry record of one (nx, nu, N) instantiation of admm_solve_kernel.
#pragma once
#include "admm_kernel.hip.h"

namespace tinympc_amd {
typedef void (*SolveKernel)(const SolveArgs);
struct KernelEntry {
    int nx, nu, N;
    SolveKernel k[2][2][3];   // [soc][dbg][dpp_mode]
    SolveKernel klin[2][4];   // [soc][LIN 1..3]: register-resident linear constraints (dpp_mode 2, no debug outputs)
    SolveKernel khet[2];      // [soc]: per-instance problem data (dpp_mode 2, no debug outputs)
    SolveKernel khetub[2];    // [soc]: the same with the knot-invariant box in registers (round 5: +4 % on the hover episode); nullptr: instantiated at run time
    SolveKernel kadapt[2];    // [dbg]: adaptive rho (dpp_mode 2, no cone)
    SolveKernel kub;          // knot-invariant box in registers (plain variant, dpp_mode 2)
    SolveKernel kubsoc;       // the same for the cone variant (its slack lives in LDS since round 4: the two bound registers fit)
    SolveKernel khalf[2];     // [UB]: HALF rows -- nx+nu <= 8, two instances per DPP row, eight per wave (plain box variant; else nullptr)
    SolveKernel kpf[2];       // [UB]: the PREFETCH form of the plain box variant (round 6: persistent waves, the next tile's records by LDS-DMA); nullptr: none
    SolveKernel khalfpf[2];   // [UB]: ... of the HALF form
};
// the PREFETCH form exists where the tile buffer of a wave (x0 piece + up to four record arrays) fits the LDS share of the waves the
// variant runs per CU, and a 16-byte piece never straddles two instances' records (N * (nx+nu) even)
constexpr bool pf_shape(int nx, int nu, int n, bool half) {
    const int nz = nx + nu, ipw = half ? 8 : 4;
    const long arr = ((long)ipw * n * nz * 8 + 1023) / 1024 * 1024;
    const int waves = 4 * solve_kernel_waves_per_simd(nz, n, false);
    // (two waves per SIMD only: the one long-horizon shape whose buffer would fit, (8,2,30), hung in its ticketed tiles on the GPU --
    // round 6, tools/experiments/prefetch_hang_8_2_30.py -- and a solve of hundreds of iterations has nothing to gain from the form)
    if (waves < 8) return false;
    return fused_shape(nx, nu) && (n * nz) % 2 == 0 && 1024 + 3 * arr + 8L * (nx * 16 + 2 * n * 16) <= 160L * 1024 / waves;
}
template <int NX, int NU, int NN, bool UB, bool HALF>
constexpr SolveKernel pf_kernel_or_null() {
    if constexpr (pf_shape(NX, NU, NN, HALF) && (!HALF || NX + NU <= 8)) return admm_solve_kernel<NX, NU, NN, false, false, 2, 0, false, LIN_KMAX, false, UB, HALF, true>;
    else return nullptr;
}
template <int NX, int NU, int NN, bool UB>
constexpr SolveKernel half_kernel_or_null() {
    if constexpr (NX + NU <= 8 && fused_shape(NX, NU)) return admm_solve_kernel<NX, NU, NN, false, false, 2, 0, false, LIN_KMAX, false, UB, true>;
    else return nullptr;
}
struct TileEntry {
    int nx, nu, N, W, R;
    int lm;                   // which arrays leave the register file (tile_kernel.hip.h TILE_LM_*); 99: the largest set that fits the wave's LDS share
    SolveKernel k;            // box table in LDS (nullptr when the wave's LDS would not hold it next to the offloaded arrays)
    SolveKernel kub;          // knot-invariant box in registers (nullptr for run-time instantiated tile shapes)
    SolveKernel kdyn, kubdyn; // the same two forms on a persistent grid whose slots draw instances from a device-wide counter
};
}  // namespace tinympc_amd

#define KERNELS_MODES(NX, NU, NN, S, D)                                                          \
    { tinympc_amd::admm_solve_kernel<NX, NU, NN, S, D, 0>, tinympc_amd::admm_solve_kernel<NX, NU, NN, S, D, 1>, \
      tinympc_amd::admm_solve_kernel<NX, NU, NN, S, D, 2> }
#define KERNELS_LIN(NX, NU, NN, S)                                                                \
    { nullptr, tinympc_amd::admm_solve_kernel<NX, NU, NN, S, false, 2, 1>,                         \
      tinympc_amd::admm_solve_kernel<NX, NU, NN, S, false, 2, 2>, tinympc_amd::admm_solve_kernel<NX, NU, NN, S, false, 2, 3> }
#define KERNELS_FOR(NX, NU, NN)                                                                   \
    { NX, NU, NN, { { KERNELS_MODES(NX, NU, NN, false, false), KERNELS_MODES(NX, NU, NN, false, true) },   \
                    { KERNELS_MODES(NX, NU, NN, true, false), KERNELS_MODES(NX, NU, NN, true, true) } },    \
      { KERNELS_LIN(NX, NU, NN, false), KERNELS_LIN(NX, NU, NN, true) },                                    \
      { tinympc_amd::admm_solve_kernel<NX, NU, NN, false, false, 2, 0, true>,                               \
        tinympc_amd::admm_solve_kernel<NX, NU, NN, true, false, 2, 0, true> },                              \
      { tinympc_amd::admm_solve_kernel<NX, NU, NN, false, false, 2, 0, true, tinympc_amd::LIN_KMAX, false, true>, \
        tinympc_amd::admm_solve_kernel<NX, NU, NN, true, false, 2, 0, true, tinympc_amd::LIN_KMAX, false, true> }, \
      { tinympc_amd::admm_solve_kernel<NX, NU, NN, false, false, 2, 0, false, tinympc_amd::LIN_KMAX, true>, \
        tinympc_amd::admm_solve_kernel<NX, NU, NN, false, true, 2, 0, false, tinympc_amd::LIN_KMAX, true> },  \
      tinympc_amd::admm_solve_kernel<NX, NU, NN, false, false, 2, 0, false, tinympc_amd::LIN_KMAX, false, true>,         \
      tinympc_amd::admm_solve_kernel<NX, NU, NN, true, false, 2, 0, false, tinympc_amd::LIN_KMAX, false, true>,          \
      { tinympc_amd::half_kernel_or_null<NX, NU, NN, false>(), tinympc_amd::half_kernel_or_null<NX, NU, NN, true>() },           \
      { tinympc_amd::pf_kernel_or_null<NX, NU, NN, false, false>(), tinympc_amd::pf_kernel_or_null<NX, NU, NN, true, false>() }, \
      { tinympc_amd::pf_kernel_or_null<NX, NU, NN, false, true>(), tinympc_amd::pf_kernel_or_null<NX, NU, NN, true, true>() } }
// the LEAN set of a shape that only the sweep of BASELINE configs[4] asks for: the box kernel in its two bound forms; its cone /
// half-space / per-instance-data / adaptive / debug / dpp-mode variants are instantiated at run time on first use (jit.hip) --
// compiled in, every one of them cost build time and library size for launches nobody has measured
#define KERNELS_LEAN(NX, NU, NN)                                                                  \
    { NX, NU, NN, { { { nullptr, nullptr, tinympc_amd::admm_solve_kernel<NX, NU, NN, false, false, 2> }, { nullptr, nullptr, nullptr } },   \
                    { { nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr } } },                       \
      { { nullptr, nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr, nullptr } },                     \
      { nullptr, nullptr }, { nullptr, nullptr }, { nullptr, nullptr },                                       \
      tinympc_amd::admm_solve_kernel<NX, NU, NN, false, false, 2, 0, false, tinympc_amd::LIN_KMAX, false, true>, nullptr,  \
      { tinympc_amd::half_kernel_or_null<NX, NU, NN, false>(), tinympc_amd::half_kernel_or_null<NX, NU, NN, true>() },           \
      { tinympc_amd::pf_kernel_or_null<NX, NU, NN, false, false>(), tinympc_amd::pf_kernel_or_null<NX, NU, NN, true, false>() }, \
      { tinympc_amd::pf_kernel_or_null<NX, NU, NN, false, true>(), tinympc_amd::pf_kernel_or_null<NX, NU, NN, true, true>() } }
