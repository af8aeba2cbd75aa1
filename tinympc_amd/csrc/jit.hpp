// jit.hpp -- run-time instantiation of the register-resident kernel (admm_kernel.hip.h) for (nx, nu, N) shapes that
// are not in kernel_dims.txt.  The kernel header is embedded in the library at build time (_gen/kernel_src.inc) and
// compiled for gfx950 with hipRTC on first use (seconds per variant; cached per process, and across processes in the
// directory TINYMPC_AMD_JIT_CACHE names, if set); libhiprtc is
// dlopen'ed so that the library loads without it -- then such shapes simply stay on the coverage kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace tinympc_amd {

struct JitKey {
    int nx, nu, N, soc, dbg, mode, lin, het, kmax, adapt;
    int ub = 0;               // the knot-invariant-bounds form (round 5: for the per-instance-data variant, which is only compiled in without it)
    bool operator<(const JitKey& o) const {
        const int a[11] = {nx, nu, N, soc, dbg, mode, lin, het, kmax, adapt, ub}, b[11] = {o.nx, o.nu, o.N, o.soc, o.dbg, o.mode, o.lin, o.het, o.kmax, o.adapt, o.ub};
        for (int i = 0; i < 11; ++i)
            if (a[i] != b[i]) return a[i] < b[i];
        return false;
    }
};

// Does the one-row kernel hold this shape at all (one instance per 16-lane row, N-long arrays in <= 512 registers)?
inline bool jit_shape_fits(int nx, int nu, int N, bool soc) {
    return nx >= 1 && nu >= 1 && nx + nu <= 16 && N >= 2 && 2 * ((soc ? 8 : 6) * N + 2 * (nx + nu) + 8) + 40 <= 512;
}

// The compiled kernel, or nullptr (reason in *err).  Thread-safe; failures are remembered.
hipFunction_t jit_solve_kernel(const JitKey& key, std::string* err);

// Tile kernel (tile_kernel.hip.h) for wide (16 < nx+nu <= 32) and / or long shapes: W rows across the knot vector, R rows
// along the horizon.  Picks the smallest R in {1, 2, 4} with W*R in {1, 2, 4}, N % R == 0, arrays within 512 registers
// and the per-wave bound / trajectory tables within the static LDS limit; false if there is none.
inline bool jit_tile_shape(int nx, int nu, int N, int* W, int* R) {
    const int nz = nx + nu;
    if (nx < 1 || nu < 1 || nz > 32 || N < 2) return false;
    const int w = nz > 16 ? 2 : 1;
    for (int r = 1; r <= 4 / w; r *= 2) {
        if (N % r) continue;
        const long lds = 2L * N * 16 * w * 8 + (long)(N / r) * 64 * 8;
        if (2 * (5 * (N / r) + 2 * nz) + 44 <= 512 && lds <= 60 * 1024) { *W = w; *R = r; return true; }
    }
    return false;
}
// soc: the tile kernel's SOC template value -- bit 0 the input family's cone slack is on, bit 1 the state family's
// ub: the UB form (the box is the same at every knot: a lane's bounds in registers) -- static form only
// ext: the tile kernel's EXT template value -- bit 0 reference window / reset_duals / cold start / store masks, bit 1 per-instance problem data
// lm (ext == 2 only): the TILE_LM_* set of the box form the per-instance data rides on
hipFunction_t jit_tile_kernel(int nx, int nu, int N, int W, int R, int soc, int lin, int kmax, std::string* err, bool dyn = false, bool ub = false, int ext = 0, int lm = 0);

// Compile one instantiation ("tinympc_amd::admm_solve_kernel<...>" / "tinympc_amd::admm_tile_kernel<...>") without loading
// it -- needs no GPU.  With TINYMPC_AMD_JIT_CACHE=<directory> set the code object is looked up / kept there (one file per
// instantiation, keyed by the kernel sources, the options and the hipRTC version; written atomically).  Returns the code
// size in bytes, -1 on failure (*err); *from_disk = 1 when nothing had to be compiled.
long jit_compile_only(const char* instantiation, int* from_disk, std::string* err);
// build time: compile with this process's hipRTC and keep the code object in the prebuilt store `dir` (null: next to the library); 0: already there
long jit_prebuild(const char* instantiation, const char* dir, std::string* err);
int jit_used_names(std::string* out);       // the instantiations compiled / loaded so far, one per line; returns their number

}  // namespace tinympc_amd
