#include <hip/hip_runtime.h>
#include <cstdio>
// which spelling / sequence of bank-masked DP-ALU DPP FMAs takes effect?
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    double src = (double)lane, m = 2.0;
    double a = 100.0 + lane;      // A: literal lane, single
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n" : "+&v"(a) : "v"(src), "v"(m));
    out[lane] = a;
    double b = 100.0 + lane;      // B: expression lane, single
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:0+3 row_mask:0xf bank_mask:0x3\n" : "+&v"(b) : "v"(src), "v"(m));
    out[64 + lane] = b;
    double c = 100.0 + lane;      // C: pair, literal lanes
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc\n" : "+&v"(c) : "v"(src), "v"(m));
    out[128 + lane] = c;
    double d = 100.0 + lane;      // D: pair, non-volatile asm
    asm("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc\n" : "+&v"(d) : "v"(src), "v"(m));
    out[192 + lane] = d;
    double e = 100.0 + lane;      // E: pair with expressions as the kernel spells them
    asm("s_nop 1\n\t v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3+3 row_mask:0xf bank_mask:0x3\n\t v_fmac_f64_dpp %0, %1, %2 row_newbcast:8+%3+3 row_mask:0xf bank_mask:0xc\n\t" : "+&v"(e) : "v"(src), "v"(m), "i"(0));
    out[256 + lane] = e;
    double g = 100.0 + lane;      // G: all low-half FMAs, two wait states, all high-half FMAs
    asm("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0x3\n"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0x3\n s_nop 1\n"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xc\n"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:13 row_mask:0xf bank_mask:0xc\n" : "+&v"(g) : "v"(src), "v"(m));
    out[384 + lane] = g;
    double h2 = 100.0 + lane;     // H: pair with two wait states in between
    asm("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc\n" : "+&v"(h2) : "v"(src), "v"(m));
    out[448 + lane] = h2;
    double i2 = 100.0 + lane;     // I: pair with ONE wait state in between
    asm("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n s_nop 0\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc\n" : "+&v"(i2) : "v"(src), "v"(m));
    out[512 + lane] = i2;
    double f = 100.0;             // F: acc uniform (as in the failing test): pair literal
    asm("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc\n" : "+&v"(f) : "v"(src), "v"(m));
    out[320 + lane] = f;
}
int main() {
    double* d; hipMalloc(&d, 1024 * 8);
    probe<<<1, 64>>>(d);
    double h[576]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"A literal single", "B expression single", "C pair literal volatile", "D pair literal", "E pair expressions", "F pair literal, uniform acc"};
    for (int t = 0; t < 6; ++t) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int row = l / 16, j = l % 16;
            double e = (t == 5 ? 100.0 : 100.0 + l);
            if (j < 8) e += 2.0 * (row * 16 + 3);
            else if (t >= 2) e += 2.0 * (row * 16 + 11);
            if (h[t * 64 + l] != e) { if (bad < 2) printf("   %s lane %d: %.1f expected %.1f\n", names[t], l, h[t * 64 + l], e); ++bad; }
        }
        printf("%-28s %s\n", names[t], bad ? "WRONG" : "ok");
    }
    for (int t = 0; t < 3; ++t) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int row = l / 16, j = l % 16;
            double e = 100.0 + l;
            if (t == 0) e += (j < 8) ? 2.0 * (3 * row * 16 + 3 + 4 + 5) : 2.0 * (3 * row * 16 + 11 + 12 + 13);
            else e += (j < 8) ? 2.0 * (row * 16 + 3) : 2.0 * (row * 16 + 11);
            if (h[384 + t * 64 + l] != e) { if (bad < 2) printf("   lane %d: %.1f expected %.1f\n", l, h[384 + t * 64 + l], e); ++bad; }
        }
        printf("%-28s %s\n", t == 0 ? "G low chain, nop 1, high chain" : (t == 1 ? "H pair, s_nop 1 between" : "I pair, s_nop 0 between"), bad ? "WRONG" : "ok");
    }
    return 0;
}
