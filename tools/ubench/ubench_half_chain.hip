#include <hip/hip_runtime.h>
#include <cstdio>
// Latency of a sweep step's FMA chain when ONE wave owns the SIMD (the tile kernel's long-horizon forms): 6 columns (nx+nu = 6), each step's
// source vector is the previous step's result (as in the sweeps).  A: full row, one accumulator.  B: half rows as shipped in round 3
// (all low-half FMAs, s_nop 0, all high-half FMAs, ONE accumulator: 12 dependent FMAs).  C: half rows, TWO accumulators (low / high
// halves interleaved: no dependent neighbours, no wait state for the bank-mask read), merged by a select (two v_cndmask_b32).
#define LO(k) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0x3\n\t"
#define HI(k) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8+" #k " row_mask:0xf bank_mask:0xc\n\t"
#define FU(k) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
#define LO2(k) "v_fmac_f64_dpp %0, %2, %3 row_newbcast:" #k " row_mask:0xf bank_mask:0x3\n\t"
#define HI2(k) "v_fmac_f64_dpp %1, %2, %3 row_newbcast:8+" #k " row_mask:0xf bank_mask:0xc\n\t"
__global__ void probe(double* out, long long* clk, int n) {
    const int lane = threadIdx.x;
    const double m = 0.125;
    const bool hi = (lane & 8) != 0;
    double src = 1.0 + lane * 0.001;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        double a = 0.5;
        asm volatile("s_nop 1\n\t" FU(0) FU(1) FU(2) FU(3) FU(4) FU(5) : "+&v"(a) : "v"(src), "v"(m));
        src = a;
    }
    long long t1 = wall_clock64();
    out[lane] = src;
    src = 1.0 + lane * 0.001;
    for (int i = 0; i < n; ++i) {
        double a = 0.5;
        asm volatile("s_nop 1\n\t" LO(0) LO(1) LO(2) LO(3) LO(4) LO(5) "s_nop 0\n\t" HI(0) HI(1) HI(2) HI(3) HI(4) HI(5) : "+&v"(a) : "v"(src), "v"(m));
        src = a;
    }
    long long t2 = wall_clock64();
    out[64 + lane] = src;
    src = 1.0 + lane * 0.001;
    for (int i = 0; i < n; ++i) {
        double a = 0.5, b = 0.5;
        asm volatile("s_nop 1\n\t" LO2(0) HI2(0) LO2(1) HI2(1) LO2(2) HI2(2) LO2(3) HI2(3) LO2(4) HI2(4) LO2(5) HI2(5)
                     : "+&v"(a), "+&v"(b) : "v"(src), "v"(m));
        src = hi ? b : a;
    }
    long long t3 = wall_clock64();
    out[128 + lane] = src;
    // D: 12 INDEPENDENT plain v_fma_f64 (no DPP); E: 12 independent DPP FMAs; F: 12 independent v_fma_f32 -- what one wave alone can issue
    double r[12];
    for (int k = 0; k < 12; ++k) r[k] = 1.0 + k + lane;
    long long t4 = wall_clock64();
    for (int i = 0; i < n; ++i)
        asm volatile("v_fma_f64 %0, %0, %12, %13\n\tv_fma_f64 %1, %1, %12, %13\n\tv_fma_f64 %2, %2, %12, %13\n\tv_fma_f64 %3, %3, %12, %13\n\t"
                     "v_fma_f64 %4, %4, %12, %13\n\tv_fma_f64 %5, %5, %12, %13\n\tv_fma_f64 %6, %6, %12, %13\n\tv_fma_f64 %7, %7, %12, %13\n\t"
                     "v_fma_f64 %8, %8, %12, %13\n\tv_fma_f64 %9, %9, %12, %13\n\tv_fma_f64 %10, %10, %12, %13\n\tv_fma_f64 %11, %11, %12, %13\n\t"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11])
                     : "v"(m), "v"(src));
    long long t5 = wall_clock64();
    for (int i = 0; i < n; ++i)
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %13, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %13, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %13, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %13, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %13, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %13, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %6, %13, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %13, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %8, %13, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %9, %13, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %10, %13, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %11, %13, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11])
                     : "v"(m), "v"(src));
    long long t6 = wall_clock64();
    float f[12];
    for (int k = 0; k < 12; ++k) f[k] = 1.0f + k + lane;
    const float mf = 0.125f, sf = (float)src;
    for (int i = 0; i < n; ++i)
        asm volatile("v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\t"
                     "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13\n\t"
                     "v_fma_f32 %8, %8, %12, %13\n\tv_fma_f32 %9, %9, %12, %13\n\tv_fma_f32 %10, %10, %12, %13\n\tv_fma_f32 %11, %11, %12, %13\n\t"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11])
                     : "v"(mf), "v"(sf));
    long long t7 = wall_clock64();
    double acc = 0; for (int k = 0; k < 12; ++k) acc += r[k] + f[k];
    out[192 + lane] = acc;
    if (lane == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; clk[2] = t3 - t2; clk[3] = t5 - t4; clk[4] = t6 - t5; clk[5] = t7 - t6; }
}
int main() {
    double* d; long long* c; hipMalloc(&d, 256 * 8); hipMalloc(&c, 64);
    const int n = 100000;
    probe<<<1, 64>>>(d, c, n);
    probe<<<1, 64>>>(d, c, n);
    double h[192]; long long hc[6];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) if (h[64 + l] != h[128 + l]) ++bad;
    const char* nm[3] = {"A full row, 6 FMAs, one accumulator", "B half rows, 6+6 FMAs, one accumulator (round 3)", "C half rows, 6+6 FMAs, two accumulators + merge"};
    for (int t = 0; t < 3; ++t) printf("%-52s %.1f ns per step = %.0f cycles at 2.4 GHz\n", nm[t], hc[t] * 10.0 / n, hc[t] * 10.0 / n * 2.4);
    const char* nm2[3] = {"D 12 independent v_fma_f64", "E 12 independent v_fmac_f64_dpp", "F 12 independent v_fma_f32"};
    for (int t = 0; t < 3; ++t) printf("%-52s %.1f cycles per instruction, one wave on the SIMD\n", nm2[t], hc[3 + t] * 10.0 / n * 2.4 / 12);
    printf("B and C agree on every lane: %s (%d lanes differ)\n", bad ? "NO" : "yes", bad);
    return 0;
}
