// Semantics probe (gfx950): does the DP-ALU DPP form of an FP64 FMA honour bank_mask (write-disable per bank of 4 lanes)?
//   acc(lane) = 100 + lane;  src(lane) = lane;  m = 2
//   v_fmac_f64_dpp acc, src, m row_newbcast:9 bank_mask:0xc   -> expected: lanes 8..15 of every row: acc += src[row*16+9] * 2, lanes 0..7 untouched
//   v_fmac_f64_dpp acc, src, m row_newbcast:3 bank_mask:0x3   -> expected: lanes 0..7: acc += src[row*16+3] * 2
// Also the cost of the masked form (cycles per instruction, dependent chain) next to the unmasked one.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    double acc = 100.0 + lane, src = (double)lane, m = 2.0;
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xc\n" : "+&v"(acc) : "v"(src), "v"(m));
    out[lane] = acc;
    double acc2 = 100.0 + lane;
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n" : "+&v"(acc2) : "v"(src), "v"(m));
    out[64 + lane] = acc2;
}
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int MASKED>
__global__ void cost(double* out, long long* cyc, int iters) {
    double a0 = threadIdx.x * 1e-3, s = 1.0 + threadIdx.x * 1e-9, m = 1.0000001;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MASKED) asm volatile("s_nop 1\n" REP64("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0x3\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xc\n") : "+&v"(a0) : "v"(s), "v"(m));
        else asm volatile("s_nop 1\n" REP64("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf\n") : "+&v"(a0) : "v"(s), "v"(m));
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* d; long long* c;
    hipMalloc(&d, 1024 * sizeof(double)); hipMalloc(&c, 8);
    probe<<<1, 64>>>(d);
    double h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l / 16, j = l % 16;
        const double e1 = 100.0 + l + (j >= 8 ? 2.0 * (row * 16 + 9) : 0.0), e2 = 100.0 + l + (j < 8 ? 2.0 * (row * 16 + 3) : 0.0);
        if (h[l] != e1 || h[64 + l] != e2) { if (bad < 6) printf("lane %d: got %.1f / %.1f expected %.1f / %.1f\n", l, h[l], h[64 + l], e1, e2); ++bad; }
    }
    printf("bank_mask on v_fmac_f64_dpp row_newbcast: %s (%d lanes differ)\n", bad ? "NOT as expected" : "honoured: masked banks keep their value", bad);
    for (int masked = 0; masked < 2; ++masked) {
        long long cy;
        if (masked) cost<1><<<1, 64>>>(d, c, 200); else cost<0><<<1, 64>>>(d, c, 200);
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        printf("%s chain: %.2f cycles per v_fmac_f64_dpp (1 wave)\n", masked ? "bank-masked" : "unmasked", (double)cy / (200.0 * 128));
    }
    return 0;
}
