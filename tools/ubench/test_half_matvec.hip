#include "../../tinympc_amd/csrc/kernel_entry.hpp"
#include "../../tinympc_amd/csrc/tile_kernel.hip.h"
#include <cstdio>
using namespace tinympc_amd;
__global__ void k(double* out) {
    const int lane = threadIdx.x, j = lane & 7;
    double src = 1.0 + lane;
    double m[6];
    for (int c = 0; c < 6; ++c) m[c] = 0.5 * (j + 1) + c;
    out[lane] = tile_matvec<0, 0, 6>(100.0, src, m);
    double m2[2] = {2.0 + j, 3.0 + j};
    out[64 + lane] = tile_matvec<0, 4, 6>(7.0, src, m2);
}
int main() {
    double* d; hipMalloc(&d, 128 * 8);
    k<<<1, 64>>>(d);
    double h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int j = l & 7, base = l & ~7;
        double e = 100.0, e2 = 7.0;
        for (int c = 0; c < 6; ++c) e += (1.0 + base + c) * (0.5 * (j + 1) + c);
        e2 += (1.0 + base + 4) * (2.0 + j) + (1.0 + base + 5) * (3.0 + j);
        if (h[l] != e || h[64 + l] != e2) { if (bad < 8) printf("lane %d: %.2f (exp %.2f)  %.2f (exp %.2f)\n", l, h[l], e, h[64 + l], e2); ++bad; }
    }
    printf("half-row matvec: %d lanes wrong\n", bad);
    return 0;
}
