// kernel_entry.hpp -- registry record of one (nx, nu, N) instantiation of admm_solve_kernel.
#pragma once
#include "admm_kernel.hip.h"

namespace tinympc_amd {
typedef void (*SolveKernel)(const SolveArgs);
struct KernelEntry {
    int nx, nu, N;
    SolveKernel k[2][2][3];   // [soc][dbg][dpp_mode]
};
struct TileEntry {
    int nx, nu, N, W, R;
    SolveKernel k;
};
}  // namespace tinympc_amd

#define KERNELS_MODES(NX, NU, NN, S, D)                                                          \
    { tinympc_amd::admm_solve_kernel<NX, NU, NN, S, D, 0>, tinympc_amd::admm_solve_kernel<NX, NU, NN, S, D, 1>, \
      tinympc_amd::admm_solve_kernel<NX, NU, NN, S, D, 2> }
#define KERNELS_FOR(NX, NU, NN)                                                                   \
    { NX, NU, NN, { { KERNELS_MODES(NX, NU, NN, false, false), KERNELS_MODES(NX, NU, NN, false, true) },   \
                    { KERNELS_MODES(NX, NU, NN, true, false), KERNELS_MODES(NX, NU, NN, true, true) } } }
