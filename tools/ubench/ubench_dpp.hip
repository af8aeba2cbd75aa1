// Micro-benchmark: issue cost / dependent latency of the FP64 VALU ops the ADMM kernel is made of
// (gfx950).  One block; blockDim = 64 -> 1 wave on one SIMD, 512 -> 2 waves per SIMD.
// Prints shader cycles (s_memtime) per instruction, measured on wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int TEST>
__global__ void k(double* out, long long* cyc, int iters) {
    double a0 = threadIdx.x * 1e-3, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4.0, a5 = 5.0, a6 = 6.0, a7 = 7.0;
    double s = 1.0 + threadIdx.x * 1e-9, m = 1.0000001;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (TEST == 0) {   // v_fma_f64 independent x8
            asm volatile(REP16("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                               "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
        } else if constexpr (TEST == 1) {   // v_fma_f64 dependent chain
            asm volatile(REP64("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n") : "+v"(a0) : "v"(s), "v"(m));
        } else if constexpr (TEST == 2) {   // v_fmac_f64_dpp independent x8
            asm volatile("s_nop 1\n" REP16("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n")
                         : "+&v"(a0), "+&v"(a1), "+&v"(a2), "+&v"(a3), "+&v"(a4), "+&v"(a5), "+&v"(a6), "+&v"(a7) : "v"(s), "v"(m));
        } else if constexpr (TEST == 3) {   // v_fmac_f64_dpp dependent single chain
            asm volatile("s_nop 1\n" REP64("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n")
                         : "+&v"(a0) : "v"(s), "v"(m));
        } else if constexpr (TEST == 4) {   // v_fmac_f64_dpp two chains alternating
            asm volatile("s_nop 1\n" REP64("v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n")
                         : "+&v"(a0), "+&v"(a1) : "v"(s), "v"(m));
        } else if constexpr (TEST == 5) {   // v_fmac_f64 (no dpp) two chains
            asm volatile(REP64("v_fmac_f64 %0, %2, %3\n v_fmac_f64 %1, %2, %3\n") : "+v"(a0), "+v"(a1) : "v"(s), "v"(m));
        } else if constexpr (TEST == 6) {   // v_add_f64 independent x8
            asm volatile(REP16("v_add_f64 %0, %8, %0\n v_add_f64 %1, %8, %1\n v_add_f64 %2, %8, %2\n v_add_f64 %3, %8, %3\n"
                               "v_add_f64 %4, %8, %4\n v_add_f64 %5, %8, %5\n v_add_f64 %6, %8, %6\n v_add_f64 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
        } else if constexpr (TEST == 7) {   // v_mov_b64 independent
            asm volatile(REP16("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
        } else if constexpr (TEST == 8) {   // v_max_f64 dependent chain (the residual maxima)
            asm volatile(REP64("v_max_f64 %0, %0, |%1|\n v_max_f64 %0, %0, |%2|\n") : "+v"(a0) : "v"(s), "v"(m));
        } else if constexpr (TEST == 9) {   // the real backward step shape: 12 + 4 fmac_dpp on two chains + s_nop + add + fma + mul
            asm volatile(REP4("s_nop 1\n"
                REP4("v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n")
                "v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n"
                "v_fmac_f64_dpp %0, %3, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %3, %2 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %3, %2 row_newbcast:14 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %3, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
                "v_add_f64 %0, %0, %1\n v_mul_f64 %1, %0, %3\n")
                         : "+&v"(a0), "+&v"(a1) : "v"(s), "v"(m));
        } else if constexpr (TEST == 10 || TEST == 11 || TEST == 12) {
            // what does the leading "s_nop 1" of a DPP block cost?  16-FMA single-chain blocks (the sweep's shape), the chain's
            // result being the next block's DPP source: 10 = with the nop (as shipped), 11 = block source not written in between
            // (no hazard, no nop: the floor), 12 = two independent FP64 adds in the nop's place (the hazard's wait states filled
            // with useful work)
#define BLK16(SRC) REP16("v_fmac_f64_dpp %0, " SRC ", %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
            if constexpr (TEST == 10)
                asm volatile(REP4("s_nop 1\n" BLK16("%1") "v_mov_b64 %1, %0\n") : "+&v"(a0), "+&v"(a1) : "v"(s), "v"(m));
            else if constexpr (TEST == 11)
                asm volatile(REP4(BLK16("%2") "v_mov_b64 %1, %0\n") : "+&v"(a0), "+&v"(a1) : "v"(s), "v"(m));
            else
                asm volatile(REP4("v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                                  REP16("v_fmac_f64_dpp %0, %1, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n") "v_mov_b64 %1, %0\n")
                             : "+&v"(a0), "+&v"(a1), "+&v"(a2), "+&v"(a3) : "v"(s), "v"(m));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <int TEST>
void run(const char* name, int ninstr_per_iter) {
    double* out; long long* cyc;
    hipMalloc(&out, 512 * sizeof(double)); hipMalloc(&cyc, sizeof(long long));
    for (int threads : {64, 256, 512}) {
        const int iters = 2000;
        k<TEST><<<1, threads>>>(out, cyc, 10);
        k<TEST><<<1, threads>>>(out, cyc, iters);
        long long c = 0; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-44s threads %3d (%d wave/SIMD): %6.2f cycles/instr (s_memtime ticks)\n", name, threads, threads <= 256 ? 1 : 2,
               (double)c / ((double)iters * ninstr_per_iter));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f64 independent x8", 128);
    run<1>("v_fma_f64 dependent chain", 128);
    run<2>("v_fmac_f64_dpp independent x8", 128);
    run<3>("v_fmac_f64_dpp dependent chain", 128);
    run<4>("v_fmac_f64_dpp two chains", 128);
    run<5>("v_fmac_f64 (no dpp) two chains", 128);
    run<6>("v_add_f64 independent x8", 128);
    run<7>("v_mov_b64 independent x8", 128);
    run<8>("v_max_f64 |abs| dependent chain", 128);
    run<9>("backward-step shape (18 FP64 + 2 s_nop)", 4 * 20);
    run<10>("16-FMA DPP block + mov, s_nop 1 in front (cycles per block)", 4);
    run<11>("16-FMA DPP block + mov, no hazard, no nop (cycles per block)", 4);
    run<12>("16-FMA DPP block + mov, 2 adds instead of the nop (cycles per block)", 4);
    return 0;
}
