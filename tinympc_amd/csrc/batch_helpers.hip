// batch_helpers.hip -- what the clock-checked dispatch needs besides the solve kernels: the iteration histogram of a solve and the cost
// model of the split solve (host arithmetic), the counting sorts behind step_regroup / repack_sort and the lock-step estimate (small
// device kernels of this unit), the per-instance state of adaptive rho, and the buffers all of them allocate on demand.  Split off
// batch_dispatch.hip in round 6; declarations: batch_dispatch.hpp.
#include "batch_impl.hpp"
#include "batch_dispatch.hpp"

#include <algorithm>
#include <cstring>
#include <limits>

namespace tinympc_amd {

// ---- automatic split solves ("repack_after" = -1): histogram of the iteration counts + a cost model ----------------------
static __global__ __launch_bounds__(256) void iter_hist_kernel(const int4* __restrict__ status, int batch, unsigned* __restrict__ hist) {
    __shared__ unsigned h[TinyBatch::HIST_BINS];
    for (int e = threadIdx.x; e < TinyBatch::HIST_BINS; e += blockDim.x) h[e] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < batch; i += gridDim.x * blockDim.x) {
        int it = status[i].x;
        it = it < 0 ? 0 : (it >= TinyBatch::HIST_BINS ? TinyBatch::HIST_BINS - 1 : it);
        atomicAdd(&h[it], 1u);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TinyBatch::HIST_BINS; e += blockDim.x)
        if (h[e]) atomicAdd(&hist[e], h[e]);
}

// Predicted launch time (arbitrary units: wave-iterations per wave slot) of a solve whose instances need hist[i] iterations,
// four instances per wave in lock step (a wave runs to the slowest of its rows: E[max of 4] under independence), split at
// `cap` (0 = plain) with the stage schedule of launch_solve (cap, 2 cap, 4 cap, ... max_iter).  A stage costs its work
// spread over the wave slots, or -- when it has fewer waves than slots -- the depth of its longest wave at a lone wave's
// pace (half the paired pace), plus a launch and one record reload / store per instance it carries.
static double predicted_time(const unsigned* hist, int max_iter, int cap, int growth, double slots, double launch_iters, double reload_iters, bool dynamic = false) {
    const int M = std::min(max_iter, (int)TinyBatch::HIST_BINS - 1);
    std::vector<double> cum(M + 2, 0.0);                     // cum[i] = instances with iter <= i
    double n = 0.0;
    for (int i = 0; i <= M; ++i) { n += hist[i]; cum[i] = n; }
    if (n <= 0.0) return 0.0;
    auto stage = [&](int lo, int hi, bool first) {           // iterations lo+1 .. hi for the instances with iter > lo
        const double open = first ? n : n - cum[lo];
        if (open <= 0.0) return 0.0;
        double work = 0.0, second = 0.0;                     // E[d], E[d^2] of a wave's depth d: sum_j P(d > j), sum_j (2j+1) P(d > j)
        int depth = 0;
        for (int i = lo; i < hi; ++i) {
            const double F = (cum[i] - (first ? 0.0 : cum[lo])) / open;          // P(iter <= i | open)
            const double p = 1.0 - F * F * F * F;
            work += p;
            second += (2.0 * (i - lo) + 1.0) * p;
            if (p > 1e-12) depth = i + 1 - lo;
        }
        const double waves = open / 4.0, per_slot = waves / slots;
        // a follow-up stage that walks its list with a fixed grid stride: a slot's time is the SUM of its waves' depths, the stage
        // ends with the slowest slot (mean + 2.5 sigma of that sum); the first stage is balanced by the dispatcher, a follow-up
        // stage whose waves draw their tiles from a counter (`dynamic`) ends at most one wave's depth after the mean
        const double imbalance = first ? 0.0 : (dynamic ? std::min(0.5 * depth, 2.5 * sqrt(std::max(second - work * work, 0.0)))
                                                        : 2.5 * sqrt(std::max(per_slot, 1e-9) * std::max(second - work * work, 0.0)));
        const double t = std::max(work * per_slot + imbalance, 0.5 * depth);
        return t + launch_iters + (first ? 0.0 : reload_iters * std::max(1.0, waves / slots));
    };
    if (cap <= 0 || cap >= M) return stage(0, M, true);
    double t = stage(0, cap, true);
    for (long base = cap; base < M; base *= growth) {
        const int hi = (int)std::min<long>(M, base * growth);
        t += stage((int)base, hi, false);
        if (hi >= M) break;
    }
    return t;
}

// one wave-iteration (4 instances) of the one-row kernel in microseconds: its FLOPs at ~75 % of a SIMD's FP64 issue rate (76.8 GFLOP/s
// per SIMD), shared by the waves of the SIMD -- 1.7 us for the quadrotor at two waves (measured 1.64)
double wave_iteration_us(int nx, int nu, int N, int wps) {
    const double S = (double)nx * N + (double)nu * (N - 1);
    const double fl = 4.0 * S + 2.0 * nx * nx + 3.0 * nx + (N - 1.0) * (4.0 * nx * nx + 8.0 * nx * nu + 2.0 * nu * nu + 4.0 * nu + 5.0 * nx) + 11.0 * S;
    return 4.0 * fl * wps / (76.8e3 * (wps == 2 ? 0.75 : 0.45));
}
// the K (multiple of check_termination) with the smallest predicted time, 0 when a plain launch is within 5 % of it
int choose_split_for(int nx, int nu, int N, bool soc, int M, int ct, int gr, int num_cus, const unsigned* hist, double* ratio, int* growth_out) {
    const int wps = solve_kernel_waves_per_simd(nx + nu, N, soc);
    const double slots = (double)num_cus * 4.0 * wps;         // wave slots of the chip
    // one wave-iteration in microseconds (above); the fixed costs of a stage in that unit: ~8 us of launch latency, and the record
    // reload + store (2.45 us per wave for the quadrotor's 156 slots)
    const double S = (double)nx * N + (double)nu * (N - 1);
    const double t_it = wave_iteration_us(nx, nu, N, wps);
    const double launch_iters = 8.0 / t_it, reload_iters = 2.45 * (S / 156.0) / t_it;
    const double plain = predicted_time(hist, M, 0, 2, slots, launch_iters, reload_iters);
    double best = plain;
    int best_k = 0, best_gr = gr > 0 ? gr : 2;
    // gr <= 0: the stage schedule is part of the question -- K, 2K, 4K, ... or K, 4K, 16K, ... (fewer launches, deeper lock step)
    const int grs[2] = {gr > 0 ? gr : 2, gr > 0 ? gr : 4};
    for (int gi = 0; gi < (gr > 0 ? 1 : 2); ++gi) {
        for (int k = std::max(ct, 4 - 4 % ct); k <= M / 2; k += ct) {
            if (k > 64 && k % 8) continue;                    // coarser steps far out
            const double t = predicted_time(hist, M, k, grs[gi], slots, launch_iters, reload_iters);
            if (t < best) { best = t; best_k = k; best_gr = grs[gi]; }
        }
    }
    // One check interval of margin: the histogram is the LAST solve's, and being a step late costs next to nothing (measured on config 3,
    // mode 8-9 of 262 144: K = 10 ... 14 within 3 %) while being a step early sends the whole mode through a second launch (K = 9
    // +6 %, K = 8 +60 %)
    if (best_k > 0 && best_k + ct <= M / 2 && predicted_time(hist, M, best_k + ct, best_gr, slots, launch_iters, reload_iters) <= 1.01 * best) best_k += ct;
    if (ratio) *ratio = plain > 0.0 ? best / plain : 1.0;
    if (growth_out) *growth_out = best_gr;
    return (plain > 0.0 && best < 0.95 * plain) ? best_k : 0;
}
int choose_split(const TinyBatch* b, const unsigned* hist, double* ratio) {
    return choose_split_for(b->nx, b->nu, b->N, soc_active(b), b->set.max_iter, std::max(1, b->set.check_termination), b->repack_growth >= 2 ? b->repack_growth : 0,
                            b->num_cus, hist, ratio, &const_cast<TinyBatch*>(b)->auto_growth);
}

// ---- adaptive rho: per-instance cache state + the lane tables of the adaptation step -----------------------------------
static __global__ void broadcast_vec_kernel(double* __restrict__ dst, const double* __restrict__ src, long n, int per) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i % per];
}

// every instance's cache state <- the family's cache (rho, Kinf, Pinf, C1 = Quu_inv, C2 = AmBKt: tiny_api.cpp:375-376)
int adaptive_fresh_state(TinyBatch* b) {
    const int nx = b->nx, nu = b->nu;
    std::vector<double> h;
    h.push_back(b->cache.rho);
    h.insert(h.end(), b->cache.Kinf.a.begin(), b->cache.Kinf.a.end());
    h.insert(h.end(), b->cache.Pinf.a.begin(), b->cache.Pinf.a.end());
    h.insert(h.end(), b->cache.Quu_inv.a.begin(), b->cache.Quu_inv.a.end());
    h.insert(h.end(), b->cache.AmBKt.a.begin(), b->cache.AmBKt.a.end());
    double* tmp = nullptr;
    HIP_TRY(b, hipMalloc(&tmp, h.size() * sizeof(double)));
    if (hipMemcpyAsync(tmp, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, b->stream) != hipSuccess ||
        hipStreamSynchronize(b->stream) != hipSuccess) { (void)hipFree(tmp); return fail(b, TINY_ERR_HIP, "upload of the cache state failed"); }
    struct { double* dst; int per; size_t off; } parts[] = {{b->d_arho, 1, 0}, {b->d_aK, nu * nx, 1}, {b->d_aP, nx * nx, (size_t)1 + nu * nx},
                                                            {b->d_aC1, nu * nu, (size_t)1 + nu * nx + nx * nx},
                                                            {b->d_aC2, nx * nx, (size_t)1 + nu * nx + nx * nx + nu * nu}};
    for (auto& p : parts) {
        hipLaunchKernelGGL(broadcast_vec_kernel, dim3(512), dim3(256), 0, b->stream, p.dst, tmp + p.off, (long)b->batch * p.per, p.per);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(tmp); return fail(b, TINY_ERR_HIP, "broadcast of the cache state failed"); }
    }
    const hipError_t e = hipStreamSynchronize(b->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(b, TINY_ERR_HIP, "broadcast of the cache state failed");
    b->astate_fresh = true;
    return TINY_OK;
}

int ensure_adaptive(TinyBatch* b, bool need_tables) {
    const int nx = b->nx, nu = b->nu;
    const size_t B = b->batch;
    if (!b->d_arho) {
        HIP_TRY(b, hipMalloc(&b->d_arho, B * sizeof(double)));
        HIP_TRY(b, hipMalloc(&b->d_aK, B * nu * nx * sizeof(double)));
        HIP_TRY(b, hipMalloc(&b->d_aP, B * nx * nx * sizeof(double)));
        HIP_TRY(b, hipMalloc(&b->d_aC1, B * nu * nu * sizeof(double)));
        HIP_TRY(b, hipMalloc(&b->d_aC2, B * nx * nx * sizeof(double)));
        HIP_TRY(b, hipMalloc(&b->d_atab, ATAB_DOUBLES * sizeof(double)));
        if (int rc = adaptive_fresh_state(b)) return rc;
        b->atab_dirty = true;
    }
    if (b->atab_dirty && need_tables) {
        if ((int)b->dKinf.size() != nu * nx || (int)b->dPinf.size() != nx * nx)
            return fail(b, TINY_ERR_DIM, "adaptive rho is on but the sensitivity tables are not set (tiny_batch_set_sensitivity)");
        std::vector<double> t(ATAB_DOUBLES, 0.0);
        auto at = [&](int base, int k, int j) -> double& { return t[(size_t)base + k * 16 + j]; };
        for (int j = 0; j < nx; ++j) {                       // state lanes
            for (int k = 0; k < nx; ++k) at(ATAB_AT, k, j) = b->A(k, j);                        // (A' g)_j = sum_k A[k][j] g_k
            for (int k = 0; k < nu; ++k) at(ATAB_DK, k, j) = b->dKinf[k + (size_t)nu * j];      // dK[k][j]
            for (int k = 0; k < nx; ++k) at(ATAB_DP, k, j) = b->dPinf[k + (size_t)nx * j];      // dP[k][j]
            for (int k = 0; k < nx; ++k) at(ATAB_DC2, k, j) = b->dC2.empty() ? 0.0 : b->dC2[k + (size_t)nx * j];
        }
        for (int r = 0; r < nu; ++r) {                       // input lanes
            const int j = nx + r;
            for (int k = 0; k < nx; ++k) at(ATAB_AT, k, j) = b->B(k, r);                        // (B' g)_r = sum_k B[k][r] g_k
            for (int k = 0; k < nx; ++k) at(ATAB_DK, k, j) = b->dKinf[r + (size_t)nu * k];      // dK[r][k]
        }
        for (int j = 0; j < nu; ++j)                          // C1 is nu x nu: its column j is kept by lane j
            for (int k = 0; k < nu; ++k) at(ATAB_DC1, k, j) = b->dC1.empty() ? 0.0 : b->dC1[k + (size_t)nu * j];
        HIP_TRY(b, hipMemcpyAsync(b->d_atab, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        HIP_TRY(b, hipStreamSynchronize(b->stream));
        b->atab_dirty = false;
    }
    return TINY_OK;
}

// histogram of the iteration counts the solve just enqueued leaves in d_status -> pinned host memory, asynchronously (hist_ev)
int enqueue_iteration_histogram(TinyBatch* b) {
    if (!b->d_hist) {
        HIP_TRY(b, hipMalloc(&b->d_hist, TinyBatch::HIST_BINS * sizeof(unsigned)));
        HIP_TRY(b, hipHostMalloc(reinterpret_cast<void**>(&b->h_hist), TinyBatch::HIST_BINS * sizeof(unsigned), hipHostMallocDefault));
        HIP_TRY(b, hipEventCreateWithFlags(&b->hist_ev, hipEventDisableTiming));
    }
    HIP_TRY(b, hipMemsetAsync(b->d_hist, 0, TinyBatch::HIST_BINS * sizeof(unsigned), b->stream));
    hipLaunchKernelGGL(iter_hist_kernel, dim3(64), dim3(256), 0, b->stream, b->d_status, b->batch, b->d_hist);
    HIP_TRY(b, hipGetLastError());
    HIP_TRY(b, hipMemcpyAsync(b->h_hist, b->d_hist, TinyBatch::HIST_BINS * sizeof(unsigned), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipEventRecord(b->hist_ev, b->stream));
    b->hist_pending = true;
    return TINY_OK;
}

// ---- step_regroup: counting sort of the instances by the iteration count of their last solve, largest first (the longest waves
// of a stretch start first, the short ones fill its tail), and the estimate that switches it on
enum { RG_BINS = 1024 };
__device__ __forceinline__ int regroup_key(const int4 st) {
    const int it = st.x < 0 ? -st.x : st.x;
    return it < RG_BINS - 1 ? it : RG_BINS - 1;
}
// (status / perm: of the FIRST instance of the range; `first` = its number in the batch -- what perm holds)
__global__ __launch_bounds__(256) void regroup_hist_kernel(const int4* status, int batch, unsigned* bins) {
    __shared__ unsigned h[RG_BINS];
    for (int i = threadIdx.x; i < RG_BINS; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += gridDim.x * blockDim.x) atomicAdd(&h[regroup_key(status[b])], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < RG_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&bins[i], h[i]);
}
// bins[k] <- instances with a key above k (one block of RG_BINS threads): where the first instance of key k goes
__global__ __launch_bounds__(RG_BINS) void regroup_scan_kernel(unsigned* bins) {
    __shared__ unsigned s[RG_BINS];
    const int t = threadIdx.x;
    const unsigned own = bins[RG_BINS - 1 - t];
    s[t] = own;
    __syncthreads();
    for (int d = 1; d < RG_BINS; d <<= 1) {
        const unsigned v = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    bins[RG_BINS - 1 - t] = s[t] - own;
}
// (a block ranks its 1024 instances in LDS and asks the device-wide counters once per key it holds: the counts of a batch sit in a
// handful of bins, one atomic per instance on those few addresses would serialise the whole pass)
__global__ __launch_bounds__(256) void regroup_scatter_kernel(const int4* status, int batch, int first, unsigned* bins, int* perm) {
    __shared__ unsigned base[RG_BINS], rank[RG_BINS];
    for (int c0 = blockIdx.x * 1024; c0 < batch; c0 += gridDim.x * 1024) {
        for (int i = threadIdx.x; i < RG_BINS; i += 256) { base[i] = 0u; rank[i] = 0u; }
        __syncthreads();
        int key[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = c0 + e * 256 + threadIdx.x;
            key[e] = b < batch ? regroup_key(status[b]) : -1;
            if (key[e] >= 0) atomicAdd(&base[key[e]], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < RG_BINS; i += 256)
            if (base[i]) base[i] = atomicAdd(&bins[i], base[i]);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (key[e] >= 0) perm[base[key[e]] + atomicAdd(&rank[key[e]], 1u)] = first + c0 + e * 256 + threadIdx.x;
        __syncthreads();
    }
}
// ---- repack_sort: the open instances of a split solve's stage by their distance from the tolerances.  Key = the larger of
// primal residual / tol_pri and dual residual / tol_dua (d_resid: what the stage before left at its last test), 16 bins per octave
// from 2^-8 up; a residual that is not a positive number (a diverged instance) goes in front with the largest
__device__ __forceinline__ int repack_key(const double* resid, const int b, const double rtp, const double rtd) {
    const double4 r = *reinterpret_cast<const double4*>(resid + (size_t)b * 4);
    const double m = fmax(fmax(r.x, r.y) * rtp, fmax(r.z, r.w) * rtd);
    if (!(m > 0.0) || !(m < 1e300)) return RG_BINS - 1;
    const unsigned long long u = (unsigned long long)__double_as_longlong(m);
    const int k = (int)((u >> 48) & 0x7FFFull) - ((1023 - 8) << 4);        // exponent and the mantissa's top four bits
    return k < 0 ? 0 : (k > RG_BINS - 2 ? RG_BINS - 2 : k);
}
__global__ __launch_bounds__(256) void repack_hist_kernel(const int* list, const int* count, const double* resid, double rtp, double rtd, unsigned* bins) {
    __shared__ unsigned h[RG_BINS];
    for (int i = threadIdx.x; i < RG_BINS; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    const int n = *count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&h[repack_key(resid, list[i], rtp, rtd)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < RG_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&bins[i], h[i]);
}
__global__ __launch_bounds__(256) void repack_scatter_kernel(const int* list, const int* count, const double* resid, double rtp, double rtd, unsigned* bins, int* out) {
    __shared__ unsigned base[RG_BINS], rank[RG_BINS];
    const int n = *count;
    for (int c0 = blockIdx.x * 1024; c0 < n; c0 += gridDim.x * 1024) {
        for (int i = threadIdx.x; i < RG_BINS; i += 256) { base[i] = 0u; rank[i] = 0u; }
        __syncthreads();
        int key[4], inst[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = c0 + e * 256 + threadIdx.x;
            inst[e] = i < n ? list[i] : -1;
            key[e] = i < n ? repack_key(resid, inst[e], rtp, rtd) : -1;
            if (key[e] >= 0) atomicAdd(&base[key[e]], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < RG_BINS; i += 256)
            if (base[i]) base[i] = atomicAdd(&bins[i], base[i]);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (key[e] >= 0) out[base[key[e]] + atomicAdd(&rank[key[e]], 1u)] = inst[e];
        __syncthreads();
    }
}
// what lock step costs a batch whose waves take the instances four by four in their natural order, by the iteration totals each
// instance has accumulated (d_accum): out[0] += rows x the largest total of every group of four, out[1] += the totals
__global__ __launch_bounds__(256) void lockstep_estimate_kernel(const uint2* accum, int batch, unsigned long long* out) {
    unsigned long long m = 0ull, t = 0ull;
    const int groups = (batch + 3) / 4;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
        unsigned mx = 0u; int rows = 0;
        for (int r = 0; r < 4 && 4 * g + r < batch; ++r) {
            const unsigned v = accum[4 * g + r].x;
            mx = v > mx ? v : mx; t += v; ++rows;
        }
        m += (unsigned long long)mx * rows;
    }
    for (int off = 32; off >= 1; off >>= 1) { m += __shfl_xor(m, off); t += __shfl_xor(t, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], m); atomicAdd(&out[1], t); }
}
// the code object of this unit's helper kernels (histograms, counting sorts) reaches the device when one of them is first asked about:
// done where their buffers are allocated -- in front of the clock of a planned first solve -- instead of inside its first sort
static void preload_helper_kernels(TinyBatch* b) {
    if (b->helpers_loaded) return;
    b->helpers_loaded = true;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(regroup_hist_kernel)) != hipSuccess) (void)hipGetLastError();
}
int ensure_regroup_buffers(TinyBatch* b, bool second_stream) {
    preload_helper_kernels(b);
    if (!b->d_perm) HIP_TRY(b, hipMalloc(&b->d_perm, (size_t)b->batch * sizeof(int)));
    if (!b->d_rg_bins) HIP_TRY(b, hipMalloc(&b->d_rg_bins, 2 * RG_BINS * sizeof(unsigned)));
    if (second_stream && b->regroup_streams == 2 && !b->stream2) {
        HIP_TRY(b, hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking));
        HIP_TRY(b, hipEventCreateWithFlags(&b->rg_fork, hipEventDisableTiming));
        HIP_TRY(b, hipEventCreateWithFlags(&b->rg_join, hipEventDisableTiming));
    }
    return TINY_OK;
}
// d_perm[first ..) <- instances first .. first + count - 1 ordered by the iteration count d_status holds for them, largest first
// (enqueued on `st`; `half` picks the set of counters)
int enqueue_regroup_sort(TinyBatch* b, hipStream_t st, int half, int first, int count) {
    unsigned* bins = b->d_rg_bins + half * RG_BINS;
    HIP_TRY(b, hipMemsetAsync(bins, 0, RG_BINS * sizeof(unsigned), st));
    const int blocks = std::max(1, std::min(256, (count + 1023) / 1024));
    hipLaunchKernelGGL(regroup_hist_kernel, dim3(blocks), dim3(256), 0, st, b->d_status + first, count, bins);
    hipLaunchKernelGGL(regroup_scan_kernel, dim3(1), dim3(RG_BINS), 0, st, bins);
    hipLaunchKernelGGL(regroup_scatter_kernel, dim3(blocks), dim3(256), 0, st, b->d_status + first, count, first, bins, b->d_perm + first);
    HIP_TRY(b, hipGetLastError());
    return TINY_OK;
}
// d_perm <- list[0 .. *count) ordered by repack_key, largest first (on the batch's stream)
int enqueue_repack_sort(TinyBatch* b, const int* list, const int* count) {
    unsigned* bins = b->d_rg_bins;
    HIP_TRY(b, hipMemsetAsync(bins, 0, RG_BINS * sizeof(unsigned), b->stream));
    const int blocks = std::max(1, std::min(128, (b->batch + 1023) / 1024));
    const double rtp = 1.0 / b->set.abs_pri_tol, rtd = 1.0 / b->set.abs_dua_tol;
    hipLaunchKernelGGL(repack_hist_kernel, dim3(blocks), dim3(256), 0, b->stream, list, count, b->d_resid, rtp, rtd, bins);
    hipLaunchKernelGGL(regroup_scan_kernel, dim3(1), dim3(RG_BINS), 0, b->stream, bins);
    hipLaunchKernelGGL(repack_scatter_kernel, dim3(blocks), dim3(256), 0, b->stream, list, count, b->d_resid, rtp, rtd, bins, b->d_perm);
    HIP_TRY(b, hipGetLastError());
    return TINY_OK;
}
int enqueue_lockstep_estimate(TinyBatch* b) {
    if (!b->d_ls) {
        HIP_TRY(b, hipMalloc(&b->d_ls, 2 * sizeof(unsigned long long)));
        HIP_TRY(b, hipHostMalloc(reinterpret_cast<void**>(&b->h_ls), 2 * sizeof(unsigned long long), hipHostMallocDefault));
        HIP_TRY(b, hipEventCreateWithFlags(&b->ls_ev, hipEventDisableTiming));
    }
    HIP_TRY(b, hipMemsetAsync(b->d_ls, 0, 2 * sizeof(unsigned long long), b->stream));
    hipLaunchKernelGGL(lockstep_estimate_kernel, dim3(64), dim3(256), 0, b->stream, b->d_accum, b->batch, b->d_ls);
    HIP_TRY(b, hipGetLastError());
    HIP_TRY(b, hipMemcpyAsync(b->h_ls, b->d_ls, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipEventRecord(b->ls_ev, b->stream));
    b->ls_pending = true;
    return TINY_OK;
}
void read_lockstep_estimate(TinyBatch* b) {
    b->ls_pending = false;
    b->lockstep_ratio = b->h_ls[1] > 0ull ? (double)b->h_ls[0] / (double)b->h_ls[1] : 1.0;
    if (b->regroup_verdict == 0) b->regroup_verdict = b->lockstep_ratio >= 1.05 ? 1 : -1;
}
// stretches of `steps` MPC steps: `lead` steps first (0: none), then K steps each; a short remainder joins the stretch before it
std::vector<int> regroup_stretches(int steps, int K, int lead) {
    std::vector<int> out;
    int left = steps;
    if (lead > 0 && left > lead) { out.push_back(lead); left -= lead; }
    while (left > 0) {
        int n = std::min(K, left);
        if (left - n > 0 && left - n < (K + 1) / 2) n = left;
        out.push_back(n);
        left -= n;
    }
    return out;
}
int regroup_auto_k(int steps) { return std::max(8, (steps + 3) / 4); }
// the two-stream form (halves of the batch half a stretch out of step) only makes sense when more than one stretch is left after the
// lead step: the condition the launch and tiny_step_regroup_plan share
bool regroup_two_streams_apply(int steps, int lead, int K) { return steps - lead > K; }

// index lists + per-stage counters of the split solve: [stage] list lengths, [32 + stage] tile counters
int ensure_repack_buffers(TinyBatch* b) {
    if (b->d_repack_index && b->d_repack_count) return TINY_OK;
    preload_helper_kernels(b);
    if (!b->d_repack_index) HIP_TRY(b, hipMalloc(&b->d_repack_index, 2 * (size_t)b->batch * sizeof(int)));
    if (!b->d_repack_count) HIP_TRY(b, hipMalloc(&b->d_repack_count, 2 * 32 * sizeof(int)));
    return TINY_OK;
}

}  // namespace tinympc_amd
