// Micro-benchmark behind the one-row kernel's PREFETCH launch form (round 6): the record traffic of a warm solve -- R record
// arrays in, W record arrays out (one of them nontemporal), 4 instances per wave, 10 knots x 16 rows x 8 B per instance and
// array -- with a dependent FP64 chain of `chain` FMAs standing in for the ADMM iterations, at the occupancy the 247-VGPR
// kernel has (8 waves per CU) and above it.  Three launch forms:
//   0  one tile per wave (grid = tiles): what the kernel does today
//   1  persistent waves, grid stride, direct loads at the top of every tile
//   2  persistent waves + LDS-DMA (global_load_lds_dwordx4) of the NEXT tile's records while the current one computes:
//      one LDS buffer per wave (free again the moment its contents are in registers)
// Batch 262 144: every array is 320 MiB, the working set is beyond the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int N = 10, NZ = 16, REC = N * NZ;      // doubles per instance and array
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int R>
__device__ __forceinline__ void dma_tile(const double* in, size_t arr, int tile, int lane, double* buf) {
    // a tile = 4 consecutive instances = 4 * REC doubles = 5120 B per array: five 1-KiB pieces
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int q = 0; q < 5; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(in + a * arr + (size_t)tile * (4 * REC) + q * 128 + lane * 2),
                                             (lptr_t)(buf + a * (4 * REC) + q * 128), 16, 0, 0);
}

#ifndef KEEP
#define KEEP (W * N)          // stores of the tile before that may stay in flight at a tile's top (form 2); -DKEEP=20: what the solve kernel allows
#endif
#ifndef SLOT_MAJOR
#define SLOT_MAJOR 0          // 1: the write-back walks the slots and stores every array's cell of a slot (the solve kernel's order)
#endif
template <int R, int W, int MODE>
__global__ __launch_bounds__(64) void stream(const double* __restrict__ in, double* __restrict__ out, int batch, size_t arr, int chain, int reverse) {
    extern __shared__ double buf[];
    const int lane = threadIdx.x & 63, j = lane & 15, grp = lane >> 4;
    const int ntiles = batch / 4;
    auto tile_of = [&](int t) { return reverse ? ntiles - 1 - t : t; };
    int t = blockIdx.x;
    if constexpr (MODE == 2) { if (t < ntiles) dma_tile<R>(in, arr, tile_of(t), lane, buf); }
    for (; t < ntiles; t += gridDim.x) {
        const int tile = tile_of(t);
        const size_t rec = (size_t)(tile * 4 + grp) * REC;
        double v[R][N];
        if constexpr (MODE == 2) {
            // the compiler does not order a ds_read behind a pending LDS-DMA: the wave's own vmcnt does.  VMEM operations of a wave
            // complete in issue order on gfx9, and the W * N stores of the tile before were issued BEHIND this tile's DMA: they may
            // stay in flight
            if (t == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < N; ++s) v[a][s] = buf[a * (4 * REC) + grp * REC + s * NZ + j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the buffer's contents are in registers: it is free
            if (t + (int)gridDim.x < ntiles) dma_tile<R>(in, arr, tile_of(t + gridDim.x), lane, buf);
        } else {
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < N; ++s) v[a][s] = in[a * arr + rec + s * NZ + j];
        }
        double acc = v[0][N - 1] + v[1][N - 1];
        for (int i = 0; i < chain; ++i) acc = fma(acc, 1.0000001, 1e-9);
#if SLOT_MAJOR
        // SPLIT_LINES: the solve kernel's lanes -- rows 12-15 (the inputs) hold knot s-1 at slot s, so a store instruction writes 96 bytes
        // of one 128-byte line and 32 bytes of the line before it
#ifndef SPLIT_LINES
#define SPLIT_LINES 0
#endif
#pragma unroll
        for (int s = 0; s < N; ++s)
#pragma unroll
            for (int a = 0; a < W; ++a) {
                const double o = v[a % R][s] + acc;
                const size_t at = a * arr + rec + s * NZ + j - ((SPLIT_LINES && j >= 12) ? NZ : 0);
                if (SPLIT_LINES && j >= 12 && s == 0) continue;
                if (a == 0) __builtin_nontemporal_store(o, out + at);
                else out[at] = o;
            }
#else
#pragma unroll
        for (int a = 0; a < W; ++a)
#pragma unroll
            for (int s = 0; s < N; ++s) {
                const double o = v[a % R][s] + acc;
                if (a == 0) __builtin_nontemporal_store(o, out + a * arr + rec + s * NZ + j);
                else out[a * arr + rec + s * NZ + j] = o;
            }
#endif
    }
}

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 262144;
    const size_t arr = (size_t)batch * REC;
    double *in, *out;
    (void)hipMalloc(&in, 4 * arr * 8); (void)hipMalloc(&out, 4 * arr * 8);
    (void)hipMemset(in, 0, 4 * arr * 8); (void)hipMemset(out, 0, 4 * arr * 8);
#ifdef IN_PLACE
    out = in;                 // the solve kernel rewrites the records it read: vnew|znew, g|y (and v|z) in place
#endif
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int ncu = 256;
    { hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0); ncu = p.multiProcessorCount; }
    printf("batch %d, %d CUs; records %.0f MiB per array\n", batch, ncu, arr * 8 / 1048576.0);
    printf("| form | arrays in/out | waves per CU | chain | us per launch | TB/s |\n|---|---|---|---|---|---|\n");
    auto time = [&](auto launch, const char* form, int r, int w, int wpc, int chain) {
        for (int i = 0; i < 3; ++i) launch(i & 1);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch(i & 1);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("| %s | %d / %d | %d | %d | %.1f | %.2f |\n", form, r, w, wpc, chain, ms * 1e3, (double)(r + w) * arr * 8 / ms / 1e9);
        fflush(stdout);
    };
    const int ntiles = batch / 4;
    for (int chain : {0, 600, 1200}) {
        for (int wpc : {8, 12, 16}) {
            const int lds = (160 * 1024 / wpc) & ~1023;          // caps the residency at wpc waves per CU
            time([&](int rev) { stream<3, 3, 0><<<ntiles, 64, lds>>>(in, out, batch, arr, chain, rev); }, "0 one tile per wave", 3, 3, wpc, chain);
            time([&](int rev) { stream<3, 3, 1><<<ncu * wpc, 64, lds>>>(in, out, batch, arr, chain, rev); }, "1 persistent", 3, 3, wpc, chain);
            if (lds >= 3 * 4 * REC * 8)
                time([&](int rev) { stream<3, 3, 2><<<ncu * wpc, 64, lds>>>(in, out, batch, arr, chain, rev); }, "2 persistent + LDS-DMA prefetch", 3, 3, wpc, chain);
            time([&](int rev) { stream<4, 4, 0><<<ntiles, 64, lds>>>(in, out, batch, arr, chain, rev); }, "0 one tile per wave", 4, 4, wpc, chain);
            time([&](int rev) { stream<4, 4, 1><<<ncu * wpc, 64, lds>>>(in, out, batch, arr, chain, rev); }, "1 persistent", 4, 4, wpc, chain);
            if (lds >= 4 * 4 * REC * 8)
                time([&](int rev) { stream<4, 4, 2><<<ncu * wpc, 64, lds>>>(in, out, batch, arr, chain, rev); }, "2 persistent + LDS-DMA prefetch", 4, 4, wpc, chain);
        }
    }
    return 0;
}
