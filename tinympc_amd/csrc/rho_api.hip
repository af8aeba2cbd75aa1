// rho_api.hip -- the helpers of the reference's adaptive-rho module as exported symbols (rho_benchmark.hpp:43-94).
//
// Upstream these are ordinary C++ functions (no extern "C"): solve() calls benchmark_rho_adaptation every 5th iteration
// (admm.cpp:397-423) and a caller that includes rho_benchmark.hpp can call any of them on its own RhoAdapter.  Inside
// tiny_solve this library never builds their dense OSQP-style matrices -- the ADAPT kernel evaluates the same sums block by
// block -- but a program written against the header must still LINK and get the same numbers, so the
// seven symbols exist here under their Itanium names, on plain-data mirrors of the structs (an Eigen dynamic matrix is
// {double*, rows, cols} with malloc'ed storage, compat_api.hip):
//   initialize_format_matrices / format_matrices   host: they allocate and fill the caller's dense scratch matrices (data movement)
//   compute_residuals                               GPU 0: the three dense products + five max-norms in one launch
//   predict_rho / update_matrices_with_derivatives  host: a handful of scalar operations / one Taylor step of four small matrices
//   benchmark_rho_adaptation                        the four in the reference's order (with its quirk: initial_rho is read AFTER the
//                                                   cache moved, rho_benchmark.cpp:240-244)
//   micros                                          0, as upstream off-Arduino (rho_benchmark.cpp:9-11)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/tinympc_amd.h"

namespace {

void resize_zero(TinyMatrixPOD* m, int64_t r, int64_t c) {          // m = tinyMatrix::Zero(r, c): Eigen frees and re-allocates on a size change
    if (m->rows * m->cols != r * c) {
        free(m->data);
        m->data = (r * c) ? static_cast<double*>(malloc((size_t)(r * c) * sizeof(double))) : nullptr;
    }
    m->rows = r; m->cols = c;
    if (r * c) memset(m->data, 0, (size_t)(r * c) * sizeof(double));
}

// One block: y1 = A x, r_prim = y1 - z, Px = P x, ATy = A' y, r_dual = (Px + q) + ATy and the max-norms compute_residuals reports.
// A is m x n, P n x n, column-major; out5 = {max|r_prim|, max|Ax|, max|z|, max|r_dual|, max(|Px|, |ATy|, |q|)}.
__global__ __launch_bounds__(256) void rho_residuals_kernel(const double* __restrict__ A, const double* __restrict__ x, const double* __restrict__ z,
                                                            const double* __restrict__ y, const double* __restrict__ Pm, const double* __restrict__ q,
                                                            int m, int n, double* __restrict__ Ax, double* __restrict__ rprim, double* __restrict__ Px,
                                                            double* __restrict__ ATy, double* __restrict__ rdual, double* __restrict__ out5) {
    __shared__ double red[5][256];
    const int t = threadIdx.x;
    double mx[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = t; i < m; i += 256) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += A[i + (size_t)m * j] * x[j];
        const double r = s - z[i];
        Ax[i] = s; rprim[i] = r;
        mx[0] = fmax(mx[0], fabs(r)); mx[1] = fmax(mx[1], fabs(s)); mx[2] = fmax(mx[2], fabs(z[i]));
    }
    for (int j = t; j < n; j += 256) {
        double p = 0.0, a = 0.0;
        for (int k = 0; k < n; ++k) p += Pm[j + (size_t)n * k] * x[k];
        for (int i = 0; i < m; ++i) a += A[i + (size_t)m * j] * y[i];
        const double r = (p + q[j]) + a;
        Px[j] = p; ATy[j] = a; rdual[j] = r;
        mx[3] = fmax(mx[3], fabs(r));
        mx[4] = fmax(mx[4], fmax(fmax(fabs(p), fabs(a)), fabs(q[j])));
    }
    for (int k = 0; k < 5; ++k) red[k][t] = mx[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s)
            for (int k = 0; k < 5; ++k) red[k][t] = fmax(red[k][t], red[k][t + s]);
        __syncthreads();
    }
    if (t < 5) out5[t] = red[t][0];
}

// The helper runs on the device the CALLER has current (a process that also drives a TinyBatch / TinyGroup on other GPUs must
// not find its thread's current device changed by a call into this module), on a private non-blocking stream (never the
// legacy null stream, which would serialise against every other stream of the process) and with one scratch buffer per device.
struct DeviceScratch {
    std::mutex mu;
    double* d = nullptr;
    size_t doubles = 0;
    hipStream_t stream = nullptr;
    hipStream_t get_stream() {
        if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); stream = nullptr; }
        return stream;
    }
    double* get(size_t need) {
        if (need > doubles) {
            if (d) (void)hipFree(d);
            d = nullptr; doubles = 0;
            if (hipMalloc(reinterpret_cast<void**>(&d), need * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            doubles = need;
        }
        return d;
    }
};
enum { MAX_SCRATCH_DEVICES = 16 };
DeviceScratch g_scratch[MAX_SCRATCH_DEVICES];

}  // namespace

// ---- the plain-data mirrors of rho_benchmark.hpp:5-40 are TinyRhoAdapterPOD / TinyRhoBenchmarkResultPOD of include/tinympc_amd.h
static_assert(sizeof(TinyRhoAdapterPOD) == 304 && sizeof(TinyRhoBenchmarkResultPOD) == 56, "rho_benchmark.hpp:5-40 on x86-64");

extern "C" {

// the reference's C++ names (no extern "C" upstream); a reference to an Eigen matrix is a pointer to its {data, rows, cols}
uint32_t tinyamd_micros() __asm__("_Z6microsv");
void tinyamd_initialize_format_matrices(TinyRhoAdapterPOD*, int, int, int) __asm__("_Z26initialize_format_matricesP10RhoAdapteriii");
void tinyamd_format_matrices(TinyRhoAdapterPOD*, const TinyMatrixPOD*, const TinyMatrixPOD*, const TinyMatrixPOD*, const TinyMatrixPOD*,
                             const TinyMatrixPOD*, const TinyMatrixPOD*, TinyCache*, TinyWorkspace*, int)
    __asm__("_Z15format_matricesP10RhoAdapterRKN5Eigen6MatrixIdLin1ELin1ELi0ELin1ELin1EEES5_S5_S5_S5_S5_P9TinyCacheP13TinyWorkspacei");
void tinyamd_compute_residuals(TinyRhoAdapterPOD*, double*, double*, double*, double*) __asm__("_Z17compute_residualsP10RhoAdapterPdS1_S1_S1_");
double tinyamd_predict_rho(TinyRhoAdapterPOD*, double, double, double, double, double) __asm__("_Z11predict_rhoP10RhoAdapterddddd");
void tinyamd_update_matrices_with_derivatives(TinyCache*, double) __asm__("_Z32update_matrices_with_derivativesP9TinyCached");
void tinyamd_benchmark_rho_adaptation(TinyRhoAdapterPOD*, const TinyMatrixPOD*, const TinyMatrixPOD*, const TinyMatrixPOD*, const TinyMatrixPOD*,
                                      const TinyMatrixPOD*, const TinyMatrixPOD*, TinyCache*, TinyWorkspace*, int, TinyRhoBenchmarkResultPOD*)
    __asm__("_Z24benchmark_rho_adaptationP10RhoAdapterRKN5Eigen6MatrixIdLin1ELin1ELi0ELin1ELin1EEES5_S5_S5_S5_S5_P9TinyCacheP13TinyWorkspaceiP18RhoBenchmarkResult");

uint32_t tinyamd_micros() { return 0; }                                             // rho_benchmark.cpp:9-11

void tinyamd_initialize_format_matrices(TinyRhoAdapterPOD* a, int nx, int nu, int N) {      // rho_benchmark.cpp:14-43
    const int64_t n = (int64_t)nx * N + (int64_t)nu * (N - 1), m = (int64_t)(nx + nu) * (N - 1);
    resize_zero(&a->A_matrix, m, n);
    resize_zero(&a->z_vector, m, 1);
    resize_zero(&a->y_vector, m, 1);
    resize_zero(&a->x_decision, n, 1);
    resize_zero(&a->P_matrix, n, n);
    resize_zero(&a->q_vector, n, 1);
    resize_zero(&a->Ax_vector, m, 1);
    resize_zero(&a->r_prim_vector, m, 1);
    resize_zero(&a->r_dual_vector, n, 1);
    resize_zero(&a->Px_vector, n, 1);
    resize_zero(&a->ATy_vector, n, 1);
    a->format_nx = nx; a->format_nu = nu; a->format_N = N;
    a->matrices_initialized = true;
}

void tinyamd_format_matrices(TinyRhoAdapterPOD* a, const TinyMatrixPOD* x_prev, const TinyMatrixPOD* u_prev, const TinyMatrixPOD* v_prev,
                             const TinyMatrixPOD* z_prev, const TinyMatrixPOD* g_prev, const TinyMatrixPOD* y_prev, TinyCache* cache,
                             TinyWorkspace* work, int N) {                            // rho_benchmark.cpp:45-145
    if (!a->matrices_initialized) tinyamd_initialize_format_matrices(a, (int)x_prev->rows, (int)u_prev->rows, N);
    const int nx = a->format_nx, nu = a->format_nu, nz = nx + nu;
    const int64_t m = a->A_matrix.rows, n = a->A_matrix.cols;
    double* A = a->A_matrix.data;
    double* Pm = a->P_matrix.data;
    auto xcol = [&](const TinyMatrixPOD* M, int i) { return M->data + (size_t)M->rows * i; };
    // the decision vector [x_0 u_0 x_1 u_1 ... x_{N-1}] and q = [Q x_i ; R u_i] (references taken as zero, :127-143)
    for (int i = 0; i < N; ++i) {
        double* xd = a->x_decision.data + (size_t)i * nz;
        double* qd = a->q_vector.data + (size_t)i * nz;
        for (int k = 0; k < nx; ++k) { xd[k] = xcol(x_prev, i)[k]; qd[k] = work->Q.data[k] * xd[k]; }
        if (i < N - 1)
            for (int k = 0; k < nu; ++k) { xd[nx + k] = xcol(u_prev, i)[k]; qd[nx + k] = work->R.data[k] * xd[nx + k]; }
    }
    // constraint matrix: identity rows for the inputs, [A B -I] rows for the dynamics (:70-91)
    memset(A, 0, (size_t)(m * n) * sizeof(double));
    for (int i = 0; i < N - 1; ++i) {
        const int64_t c0 = (int64_t)i * nz;
        for (int k = 0; k < nu; ++k) A[((int64_t)i * nu + k) + m * (c0 + nx + k)] = 1.0;
        const int64_t r0 = (int64_t)(N - 1) * nu + (int64_t)i * nx;
        for (int c = 0; c < nx; ++c)
            for (int r = 0; r < nx; ++r) A[(r0 + r) + m * (c0 + c)] = work->Adyn.data[r + (size_t)nx * c];
        for (int c = 0; c < nu; ++c)
            for (int r = 0; r < nx; ++r) A[(r0 + r) + m * (c0 + nx + c)] = work->Bdyn.data[r + (size_t)nx * c];
        if (c0 + nz < n)
            for (int k = 0; k < nx; ++k) A[(r0 + k) + m * (c0 + nz + k)] = -1.0;
    }
    // z = [znew_i ; vnew_{i+1}], y = [y_i ; g_{i+1}] in the same row order (:94-100)
    for (int i = 0; i < N - 1; ++i) {
        for (int k = 0; k < nu; ++k) {
            a->z_vector.data[(size_t)i * nu + k] = xcol(z_prev, i)[k];
            a->y_vector.data[(size_t)i * nu + k] = xcol(y_prev, i)[k];
        }
        for (int k = 0; k < nx; ++k) {
            a->z_vector.data[(size_t)(N - 1) * nu + (size_t)i * nx + k] = xcol(v_prev, i + 1)[k];
            a->y_vector.data[(size_t)(N - 1) * nu + (size_t)i * nx + k] = xcol(g_prev, i + 1)[k];
        }
    }
    // cost matrix: diag(Q) | diag(R) blocks, Pinf on the last state (:103-124)
    memset(Pm, 0, (size_t)(n * n) * sizeof(double));
    for (int i = 0; i < N; ++i) {
        const int64_t d0 = (int64_t)i * nz;
        if (i == N - 1) {
            for (int c = 0; c < nx; ++c)
                for (int r = 0; r < nx; ++r) Pm[(d0 + r) + n * (d0 + c)] = cache->Pinf.data[r + (size_t)nx * c];
        } else {
            for (int k = 0; k < nx; ++k) Pm[(d0 + k) + n * (d0 + k)] = work->Q.data[k];
            for (int k = 0; k < nu; ++k) Pm[(d0 + nx + k) + n * (d0 + nx + k)] = work->R.data[k];
        }
    }
}

void tinyamd_compute_residuals(TinyRhoAdapterPOD* a, double* pri_res, double* dual_res, double* pri_norm, double* dual_norm) {   // :147-178
    const int64_t m = a->A_matrix.rows, n = a->A_matrix.cols;
    const double nan = std::nan("");
    auto loud = [&](const char* what) {
        fprintf(stderr, "tinympc_amd: compute_residuals: %s (this library computes on an MI355X, there is no CPU fallback)\n", what);
        *pri_res = *dual_res = *pri_norm = *dual_norm = nan;
    };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_SCRATCH_DEVICES) { (void)hipGetLastError(); return loud("no HIP device"); }
    DeviceScratch& sc = g_scratch[dev];              // the caller's current device: nothing is changed for the calling thread
    std::lock_guard<std::mutex> lk(sc.mu);
    const size_t in = (size_t)(m * n + n * n + n + 2 * m + n), out = (size_t)(2 * m + 3 * n + 5);
    double* d = sc.get(in + out);
    if (!d) return loud("hipMalloc failed");
    hipStream_t st = sc.get_stream();
    if (!st) return loud("hipStreamCreate failed");
    double *dA = d, *dP = dA + m * n, *dx = dP + n * n, *dz = dx + n, *dy = dz + m, *dq = dy + m;
    double *dAx = dq + n, *drp = dAx + m, *dPx = drp + m, *dAT = dPx + n, *drd = dAT + n, *dout = drd + n;
    bool ok = hipMemcpyAsync(dA, a->A_matrix.data, (size_t)(m * n) * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(dP, a->P_matrix.data, (size_t)(n * n) * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(dx, a->x_decision.data, (size_t)n * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(dz, a->z_vector.data, (size_t)m * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(dy, a->y_vector.data, (size_t)m * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(dq, a->q_vector.data, (size_t)n * 8, hipMemcpyHostToDevice, st) == hipSuccess;
    if (!ok) { (void)hipStreamSynchronize(st); return loud("upload failed"); }
    hipLaunchKernelGGL(rho_residuals_kernel, dim3(1), dim3(256), 0, st, dA, dx, dz, dy, dP, dq, (int)m, (int)n, dAx, drp, dPx, dAT, drd, dout);
    double o5[5];
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(a->Ax_vector.data, dAx, (size_t)m * 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(a->r_prim_vector.data, drp, (size_t)m * 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(a->Px_vector.data, dPx, (size_t)n * 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(a->ATy_vector.data, dAT, (size_t)n * 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(a->r_dual_vector.data, drd, (size_t)n * 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(o5, dout, sizeof(o5), hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
    if (!ok) { (void)hipStreamSynchronize(st); return loud("kernel or download failed"); }
    *pri_res = o5[0];
    *pri_norm = std::max(o5[1], o5[2]);
    *dual_res = o5[3];
    *dual_norm = o5[4];
}

double tinyamd_predict_rho(TinyRhoAdapterPOD* a, double pri_res, double dual_res, double pri_norm, double dual_norm, double current_rho) {   // :180-201
    const double eps = 1e-10;
    const double normalized_pri = pri_res / (pri_norm + eps);
    const double normalized_dual = dual_res / (dual_norm + eps);
    const double ratio = normalized_pri / (normalized_dual + eps);
    double new_rho = current_rho * std::sqrt(ratio);
    if (a->clip) new_rho = std::min(std::max(new_rho, a->rho_min), a->rho_max);
    return new_rho;
}

void tinyamd_update_matrices_with_derivatives(TinyCache* c, double new_rho) {       // rho_benchmark.cpp:203-217
    const double delta = new_rho - c->rho;
    auto step = [&](TinyMatrixPOD& mtx, const TinyMatrixPOD& d) {
        const int64_t k = mtx.rows * mtx.cols;
        if (d.rows * d.cols != k) return;                                            // (Eigen would assert; the tables were never initialised)
        for (int64_t i = 0; i < k; ++i) {
            const double t = delta * d.data[i];                                      // product rounded before the sum, as the x86-64 build does
            mtx.data[i] = mtx.data[i] + t;
        }
    };
    step(c->Kinf, c->dKinf_drho);
    step(c->Pinf, c->dPinf_drho);
    step(c->C1, c->dC1_drho);
    step(c->C2, c->dC2_drho);
    c->rho = new_rho;
}

void tinyamd_benchmark_rho_adaptation(TinyRhoAdapterPOD* a, const TinyMatrixPOD* x_prev, const TinyMatrixPOD* u_prev, const TinyMatrixPOD* v_prev,
                                      const TinyMatrixPOD* z_prev, const TinyMatrixPOD* g_prev, const TinyMatrixPOD* y_prev, TinyCache* cache,
                                      TinyWorkspace* work, int N, TinyRhoBenchmarkResultPOD* result) {   // rho_benchmark.cpp:219-253
    const uint32_t start = tinyamd_micros();
    tinyamd_format_matrices(a, x_prev, u_prev, v_prev, z_prev, g_prev, y_prev, cache, work, N);
    double pri_res, dual_res, pri_norm, dual_norm;
    tinyamd_compute_residuals(a, &pri_res, &dual_res, &pri_norm, &dual_norm);
    const double new_rho = tinyamd_predict_rho(a, pri_res, dual_res, pri_norm, dual_norm, cache->rho);
    tinyamd_update_matrices_with_derivatives(cache, new_rho);
    result->time_us = tinyamd_micros() - start;
    result->initial_rho = cache->rho;              // (read after the update: equals final_rho upstream too, :244)
    result->final_rho = new_rho;
    result->pri_res = pri_res; result->dual_res = dual_res;
    result->pri_norm = pri_norm; result->dual_norm = dual_norm;
}

}  // extern "C"
