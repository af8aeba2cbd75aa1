// ubench_mfma_f64.hip -- what v_mfma_f64_16x16x4_f64 costs on gfx950 in the dependency patterns a batched
// Riccati sweep would need, and how many FP64 VALU instructions hide behind it in the same wave.
//   A: 4 MFMAs accumulate one D (a 16x16 mat-vec batch, K = 16), next group starts from a fresh C   (independent groups)
//   B: same, but the next group's B operands are the previous group's D registers                  (the sweep's chain)
//   C: pattern B + F independent v_fma_f64 per MFMA (F = 4, 8, 12, 16)
// One wave per SIMD (256 blocks x 256 threads), cycles from s_memtime of wave 0 of each block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int F>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int groups, double seed) {
    const int lane = threadIdx.x & 63;
    double a0 = seed * (lane + 1), a1 = a0 * 0.5, a2 = a0 * 0.25, a3 = a0 * 0.125;     // matrix operand (constant)
    d4 d = {seed, seed * 2, seed * 3, seed * 4};
    double f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = seed + i;
    const double m = 1.0000001;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < groups; ++g) {
        d4 c = {0.0, 0.0, 0.0, 0.0};
        const double b0 = MODE == 0 ? seed : d[0], b1 = MODE == 0 ? seed : d[1], b2 = MODE == 0 ? seed : d[2], b3 = MODE == 0 ? seed : d[3];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < F; ++i) f[i & 15] = fma(f[i & 15], m, seed);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < F; ++i) f[(i + 4) & 15] = fma(f[(i + 4) & 15], m, seed);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < F; ++i) f[(i + 8) & 15] = fma(f[(i + 8) & 15], m, seed);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < F; ++i) f[(i + 12) & 15] = fma(f[(i + 12) & 15], m, seed);
        if (MODE == 0) { d[0] += c[0]; d[1] += c[1]; d[2] += c[2]; d[3] += c[3]; }     // keep the results alive, off the chain
        else d = c * 1e-3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = d[0] + d[1] + d[2] + d[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int F>
void run(const char* name) {
    const int groups = 2000, blocks = 256;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(double)); hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, F>), dim3(blocks), dim3(256), 0, 0, out, cyc, groups, 1e-9);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    // s_memtime counts at 100 MHz on gfx9 (constant clock); report both raw ticks and an estimate in shader clocks
    std::printf("%-44s %8.1f memtime ticks / group of 4 MFMA (+%2d fma each)\n", name, mean / groups, F);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0>("A independent groups");
    run<1, 0>("B D -> B operand chain");
    run<1, 4>("C chain + 4 fma per MFMA");
    run<1, 8>("C chain + 8 fma per MFMA");
    run<1, 12>("C chain + 12 fma per MFMA");
    run<1, 16>("C chain + 16 fma per MFMA");
    run<0, 16>("A independent + 16 fma per MFMA");
    // reference: the VALU alone
    return 0;
}
