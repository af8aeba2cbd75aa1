// Micro-benchmark of the record access pattern of the solve kernel (no compute):
//   A: what the kernel does today -- 8 B per lane, a 16-lane row reads one 128-B knot segment, 4 rows per wave
//      read 4 different records (1280 B apart); 4 arrays in, 4 arrays out, 10 knots.
//   B: 16 B per lane (two knots per access: a row covers 256 contiguous bytes), same bytes.
//   C: 16 B per lane, fully linear streams (the float4-copy ceiling).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 10, NZ = 16;
__global__ __launch_bounds__(64) void patA(const double* __restrict__ in, double* __restrict__ out, int batch, size_t arr) {
    const int lane = threadIdx.x & 63, j = lane & 15, grp = lane >> 4;
    for (int tile = blockIdx.x; tile * 4 < batch; tile += gridDim.x) {
        const size_t rec = (size_t)(tile * 4 + grp) * (N * NZ);
        double v[4][N];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int s = 0; s < N; ++s) v[a][s] = in[a * arr + rec + s * NZ + j];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int s = 0; s < N; ++s) out[a * arr + rec + s * NZ + j] = v[a][s] + 1.0;
    }
}
__global__ __launch_bounds__(64) void patB(const double2* __restrict__ in, double2* __restrict__ out, int batch, size_t arr2) {
    const int lane = threadIdx.x & 63, j = lane & 15, grp = lane >> 4;
    for (int tile = blockIdx.x; tile * 4 < batch; tile += gridDim.x) {
        const size_t rec = (size_t)(tile * 4 + grp) * (N * NZ / 2);
        double2 v[4][N / 2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int s = 0; s < N / 2; ++s) v[a][s] = in[a * arr2 + rec + s * NZ + j];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int s = 0; s < N / 2; ++s) { double2 t = v[a][s]; t.x += 1.0; t.y += 1.0; out[a * arr2 + rec + s * NZ + j] = t; }
    }
}
__global__ __launch_bounds__(256) void patC(const double2* __restrict__ in, double2* __restrict__ out, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 t = in[i]; t.x += 1.0; t.y += 1.0; out[i] = t;
    }
}
int main() {
    const int batch = 65536;
    const size_t arr = (size_t)batch * N * NZ, total = 4 * arr;
    double *in, *out;
    hipMalloc(&in, total * 8); hipMalloc(&out, total * 8);
    hipMemset(in, 0, total * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char* name) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-52s %8.1f us  %7.1f GB/s (read+write)\n", name, ms * 1e3, 2.0 * total * 8 / ms / 1e6);
    };
    for (int grid : {16384, 2048, 4096}) {
        char nm[96];
        snprintf(nm, 96, "A: 8 B/lane, 128-B knot segments, grid %d", grid);
        time([&] { patA<<<grid, 64>>>(in, out, batch, arr); }, nm);
        snprintf(nm, 96, "B: 16 B/lane, 256-B knot-pair segments, grid %d", grid);
        time([&] { patB<<<grid, 64>>>((const double2*)in, (double2*)out, batch, arr / 2); }, nm);
    }
    time([&] { patC<<<2048, 256>>>((const double2*)in, (double2*)out, total / 2); }, "C: 16 B/lane linear copy, 2048 x 256");
    time([&] { patC<<<8192, 256>>>((const double2*)in, (double2*)out, total / 2); }, "C: 16 B/lane linear copy, 8192 x 256");
    return 0;
}
