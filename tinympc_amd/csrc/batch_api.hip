// batch_api.hip -- host side of the batched, device-resident C ABI (include/tinympc_amd.h, part A).
//
// Mirrors the reference operator interface for the hot path (src/tinympc/tiny_api.cpp:21-147
// setup, :149-208 constraint setters, :388-411 settings, :443-477 x0/xref/uref, :384-386 solve)
// with a leading batch axis; the ADMM iteration itself is admm_kernel.hip.h.  No CPU fallback.
#include "batch_impl.hpp"
#include "batch_dispatch.hpp"

#include <cstdarg>
#include <cstdio>
#include <dlfcn.h>
#include <string>
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace tinympc_amd {

// ---- small device kernels (layout conversion, reductions) ------------------------------------
// Reference layout  : state-sized  [batch][N][nx]     input-sized [batch][N-1][nu]   (column-major matrices)
// Device KPI layout : [batch][N][nx+nu]  (knot-point interleaved, see admm_kernel.hip.h)
__global__ void pack_kpi_kernel(double* __restrict__ kpi, const double* __restrict__ src, int batch, int N,
                                int nz, int rows, int row_off, int cols, int broadcast) {
    const size_t total = (size_t)batch * cols * rows;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % rows);
        const int i = (int)((e / rows) % cols);
        const size_t b = e / ((size_t)rows * cols);
        const size_t s = broadcast ? ((size_t)i * rows + r) : e;
        kpi[(b * N + i) * nz + row_off + r] = src[s];
    }
}
__global__ void unpack_kpi_kernel(const double* __restrict__ kpi, double* __restrict__ dst, int batch, int N,
                                  int nz, int rows, int row_off, int cols) {
    const size_t total = (size_t)batch * cols * rows;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % rows);
        const int i = (int)((e / rows) % cols);
        const size_t b = e / ((size_t)rows * cols);
        dst[e] = kpi[(b * N + i) * nz + row_off + r];
    }
}
// Many fields in ONE launch (the drop-in path moves 9-20 workspace fields per solve: one launch each way instead of one per field).
// blockIdx.y = entry; an entry is a KPI field (rows x cols per instance at row_off) or, kpi == nullptr, a raw copy of `count` doubles.
struct XferEntry { double* kpi; double* raw; long off; long count; int rows, row_off, cols; };
struct XferTable { XferEntry e[40]; int n; };
__global__ void xfer_fields_kernel(const XferTable t, double* __restrict__ buf, int batch, int N, int nz, int to_device) {
    const XferEntry en = t.e[blockIdx.y];
    double* x = buf + en.off;
    if (!en.kpi) {
        for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)en.count; e += (size_t)gridDim.x * blockDim.x) {
            if (to_device) en.raw[e] = x[e]; else x[e] = en.raw[e];
        }
        return;
    }
    const size_t total = (size_t)batch * en.cols * en.rows;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % en.rows);
        const int i = (int)((e / en.rows) % en.cols);
        const size_t b = e / ((size_t)en.rows * en.cols);
        double* rec = en.kpi + (b * N + i) * nz + en.row_off + r;
        if (to_device) *rec = x[e]; else x[e] = *rec;
    }
}
__global__ void broadcast_rows_kernel(double* __restrict__ dst, const double* __restrict__ src, int batch, int n) {
    const size_t total = (size_t)batch * n;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        dst[e] = src[e % n];
}
// out[10] = {sum iter, sum solved, batch, max pri_s, max pri_i, max dua_s, max dua_i,
//            accumulated iterations, accumulated converged solves, 0}; out must be zeroed before the launch
// (non-negative doubles order like their bit patterns, so the maxima use integer atomicMax).
__global__ __launch_bounds__(256) void reduce_stats_kernel(const int4* __restrict__ status,
                                                           const double* __restrict__ resid,
                                                           const uint2* __restrict__ accum, int batch,
                                                           double* __restrict__ out) {
    double si = 0, ss = 0, ai = 0, as = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += gridDim.x * blockDim.x) {
        const int4 st = status[b];
        si += st.x;
        ss += st.y;
        const uint2 ac = accum[b];
        ai += ac.x;
        as += ac.y;
        const double4 r = *reinterpret_cast<const double4*>(resid + (size_t)b * 4);
        m0 = fmax(m0, r.x); m1 = fmax(m1, r.y); m2 = fmax(m2, r.z); m3 = fmax(m3, r.w);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        si += __shfl_xor(si, off); ss += __shfl_xor(ss, off);
        ai += __shfl_xor(ai, off); as += __shfl_xor(as, off);
        m0 = fmax(m0, __shfl_xor(m0, off)); m1 = fmax(m1, __shfl_xor(m1, off));
        m2 = fmax(m2, __shfl_xor(m2, off)); m3 = fmax(m3, __shfl_xor(m3, off));
    }
    // the block's four waves meet in LDS, so that the device-scope atomics (all on the same few addresses) stay at eight
    // per BLOCK: with one set per wave this kernel took 87 us on 65 536 instances, most of it serialised atomics
    __shared__ double part[4][8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[w][0] = si; part[w][1] = ss; part[w][2] = ai; part[w][3] = as;
        part[w][4] = m0; part[w][5] = m1; part[w][6] = m2; part[w][7] = m3;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            for (int e = 0; e < 4; ++e) part[0][e] += part[k][e];
            for (int e = 4; e < 8; ++e) part[0][e] = fmax(part[0][e], part[k][e]);
        }
        atomicAdd(out + 0, part[0][0]); atomicAdd(out + 1, part[0][1]); atomicAdd(out + 7, part[0][2]); atomicAdd(out + 8, part[0][3]);
        unsigned long long* mo = reinterpret_cast<unsigned long long*>(out);
        atomicMax(mo + 3, (unsigned long long)__double_as_longlong(part[0][4]));
        atomicMax(mo + 4, (unsigned long long)__double_as_longlong(part[0][5]));
        atomicMax(mo + 5, (unsigned long long)__double_as_longlong(part[0][6]));
        atomicMax(mo + 6, (unsigned long long)__double_as_longlong(part[0][7]));
        if (blockIdx.x == 0) out[2] = (double)batch;
    }
}

static int field_geometry(const TinyBatch* b, TinyField f, double** kpi, int* rows, int* row_off, int* cols) {
    const int nx = b->nx, nu = b->nu, N = b->N;
    const bool st = (f == TINY_F_XREF || f == TINY_F_X || f == TINY_F_VNEW || f == TINY_F_G || f == TINY_F_V ||
                     f == TINY_F_VCNEW || f == TINY_F_GC || f == TINY_F_Q || f == TINY_F_P || f == TINY_F_VLNEW ||
                     f == TINY_F_GL || f == TINY_F_VLNEW_TV || f == TINY_F_GL_TV);
    *rows = st ? nx : nu; *row_off = st ? 0 : nx; *cols = st ? N : N - 1;
    switch (f) {
        case TINY_F_XREF: case TINY_F_UREF: *kpi = b->d_ref; break;
        case TINY_F_X: case TINY_F_U: *kpi = b->d_prim; break;
        case TINY_F_VNEW: case TINY_F_ZNEW: *kpi = b->d_slack; break;
        case TINY_F_G: case TINY_F_Y: *kpi = b->d_dual; break;
        case TINY_F_V: case TINY_F_Z: *kpi = b->d_slack_prev; break;
        case TINY_F_VCNEW: case TINY_F_ZCNEW: *kpi = b->d_cslack; break;
        case TINY_F_GC: case TINY_F_YC: *kpi = b->d_cdual; break;
        case TINY_F_Q: case TINY_F_R: *kpi = b->d_dbg_qr; break;
        case TINY_F_P: case TINY_F_D: *kpi = b->d_dbg_pd; break;
        case TINY_F_VLNEW: case TINY_F_ZLNEW: *kpi = b->d_lslack; break;
        case TINY_F_GL: case TINY_F_YL: *kpi = b->d_ldual; break;
        case TINY_F_VLNEW_TV: case TINY_F_ZLNEW_TV: *kpi = b->d_tlslack; break;
        case TINY_F_GL_TV: case TINY_F_YL_TV: *kpi = b->d_tldual; break;
        default: return TINY_ERR_ARG;
    }
    return TINY_OK;
}

static int ensure_debug_buffers(TinyBatch* b) {
    if (b->d_dbg_qr) return TINY_OK;
    const size_t kpi_bytes = (size_t)b->batch * b->N * (b->nx + b->nu) * sizeof(double);
    HIP_TRY(b, hipMalloc(&b->d_dbg_qr, kpi_bytes));
    HIP_TRY(b, hipMalloc(&b->d_dbg_pd, kpi_bytes));
    HIP_TRY(b, hipMemsetAsync(b->d_dbg_qr, 0, kpi_bytes, b->stream));
    HIP_TRY(b, hipMemsetAsync(b->d_dbg_pd, 0, kpi_bytes, b->stream));
    return TINY_OK;
}

}  // namespace tinympc_amd

using namespace tinympc_amd;

extern "C" {

int tiny_batch_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tiny_batch_supported_dims(int* triples, int capacity) {
    for (int i = 0; i < g_nkernels && i < capacity; ++i) {
        triples[3 * i] = g_kernel_list[i]->nx; triples[3 * i + 1] = g_kernel_list[i]->nu; triples[3 * i + 2] = g_kernel_list[i]->N;
    }
    return g_nkernels;
}

int tiny_batch_setup(TinyBatch** out, const double* Adyn, const double* Bdyn, const double* fdyn,
                     const double* Qdiag, const double* Rdiag, double rho, int nx, int nu, int N, int batch,
                     int device, int verbose) {
    if (!out || !Adyn || !Bdyn || !Qdiag || !Rdiag) return TINY_ERR_NULL;
    *out = nullptr;
    if (nx <= 0 || nu <= 0 || N < 2 || batch <= 0) return TINY_ERR_DIM;
    if (device < 0) return TINY_ERR_ARG;
    const KernelEntry* ke = find_kernel(nx, nu, N);   // nullptr -> coverage kernel (general_kernel.hip.h)
    if (!ke && nx + nu > 32) {
        if (verbose) fprintf(stderr, "tinympc_amd: (nx,nu,N)=(%d,%d,%d): nx+nu > 32 is not supported\n", nx, nu, N);
        return TINY_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
        if (verbose) fprintf(stderr, "tinympc_amd: no usable HIP device (this library has no CPU path)\n");
        return TINY_ERR_NO_DEVICE;
    }
    TinyBatch* b = new TinyBatch();
    b->nx = nx; b->nu = nu; b->N = N; b->batch = batch; b->device = device; b->kernel = ke;
    b->tile = find_tile(nx, nu, N);
    int tw = 0, tr = 0;
    // (W,R) = (1,1) is the one-row kernel's job where that fits; the tile kernel keeps x|u in LDS and holds longer horizons
    if (!b->tile && jit_tile_shape(nx, nu, N, &tw, &tr) && (tw * tr > 1 || !jit_shape_fits(nx, nu, N, false))) {
        b->tile_dyn = {nx, nu, N, tw, tr, 0, nullptr, nullptr, nullptr, nullptr};
        b->tile = &b->tile_dyn;
        b->tile_is_jit = true;
    }
    b->A = Mat(nx, nx, Adyn); b->B = Mat(nx, nu, Bdyn);
    b->f = fdyn ? Mat(nx, 1, fdyn) : Mat(nx, 1);
    b->Qw.assign(Qdiag, Qdiag + nx); b->Rw.assign(Rdiag, Rdiag + nu);
    for (double& v : b->Qw) v += rho;                 // work->Q = diag(Q) + rho   (tiny_api.cpp:117)
    for (double& v : b->Rw) v += rho;                 // work->R = diag(R) + rho   (tiny_api.cpp:118)
    if (!precompute_cache(b->A, b->B, b->f, Mat::diag(b->Qw), Mat::diag(b->Rw), rho, &b->cache)) {   // tiny_api.cpp:136
        delete b;
        return TINY_ERR_ARG;
    }
    if (verbose && b->cache.riccati_converged) printf("Kinf converged after %d iterations\n", b->cache.riccati_iters);
    // tiny_set_default_settings, tiny_api.cpp:413-441
    b->set = Settings();
    auto bail = [&](int code) { tiny_batch_destroy(b); return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(TINY_ERR_NO_DEVICE);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(TINY_ERR_HIP);
    b->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return bail(TINY_ERR_HIP);
    b->own_stream = true;
    const int nz = nx + nu;
    const size_t kpi_bytes = (size_t)batch * N * nz * sizeof(double);
    // (+ one record and a little of zero padding: the tile kernel's VPG forms park the stores of lanes that hold no row there)
    const size_t kpi_pad = (size_t)N * nz * sizeof(double) + 1024;
    // ONE slab for the seven record arrays, each array's start SKEWED against its neighbours' (round 6): a launch touches the same
    // offset of five of them at the same time -- the records of one tile --, and arrays that hipMalloc places a multiple of a large power
    // of two apart send those accesses to the same HBM channels.  TINYMPC_KPI_SKEW (bytes, a multiple of 256; experiments) overrides the stagger
    double** kpis[] = {&b->d_ref, &b->d_prim, &b->d_slack, &b->d_dual, &b->d_slack_prev, &b->d_cslack, &b->d_cdual};
    size_t skew = KPI_SKEW_BYTES;
    if (const char* e = getenv("TINYMPC_KPI_SKEW")) skew = (size_t)strtoull(e, nullptr, 10) & ~(size_t)255;
    const size_t gap = ((kpi_bytes + kpi_pad + 255) & ~(size_t)255) + skew;
    if (getenv("TINYMPC_KPI_SEPARATE")) {             // (experiment: one allocation per array, as before round 6; leaked at destroy)
        for (int i = 0; i < 7; ++i) {
            if (hipMalloc(kpis[i], gap) != hipSuccess) return bail(TINY_ERR_HIP);
            if (hipMemsetAsync(*kpis[i], 0, gap, b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
        }
    } else {
    if (hipMalloc(&b->d_kpi_slab, 7 * gap) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemsetAsync(b->d_kpi_slab, 0, 7 * gap, b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    for (int i = 0; i < 7; ++i) *kpis[i] = (double*)((char*)b->d_kpi_slab + (size_t)i * gap);
    }
    b->stage_doubles = (size_t)batch * std::max(nx * N, nu * (N - 1));       // the largest host-layout field (nu > nx happens)
    // (+ 1 KiB: the PREFETCH form's LDS-DMA pieces are 1 KiB each and the last tile's reach past the end of x0 -- and of the record
    // arrays, whose kpi_pad covers it; nobody reads what those bytes bring)
    if (hipMalloc(&b->d_x0, (size_t)batch * nx * sizeof(double) + 1024) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemsetAsync(b->d_x0, 0, (size_t)batch * nx * sizeof(double) + 1024, b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMalloc(&b->d_pf_counter, 8 * 64) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemsetAsync(b->d_pf_counter, 0, 8 * 64, b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMalloc(&b->d_stage, b->stage_doubles * sizeof(double)) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMalloc(&b->d_status, (size_t)batch * sizeof(int4)) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemsetAsync(b->d_status, 0, (size_t)batch * sizeof(int4), b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMalloc(&b->d_resid, (size_t)batch * 4 * sizeof(double)) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemsetAsync(b->d_resid, 0, (size_t)batch * 4 * sizeof(double), b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMalloc(&b->d_stats, 10 * sizeof(double)) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMalloc(&b->d_accum, (size_t)batch * sizeof(uint2)) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemsetAsync(b->d_accum, 0, (size_t)batch * sizeof(uint2), b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    b->d_tab_doubles = tab_doubles(N, LIN_KMAX);
    if (hipMalloc(&b->d_tab, b->d_tab_doubles * sizeof(double)) != hipSuccess) return bail(TINY_ERR_HIP);
    b->tab_dirty = true;
    b->records_zero = true;                          // (every record was just zeroed)
    if (hipStreamSynchronize(b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    *out = b;
    return TINY_OK;
}

// == tiny_setup for `batch` DIFFERENT problem families: instance i has its own A_i, B_i, f_i, Q_i, R_i, rho_i.
// The Riccati recursion (tiny_api.cpp:307-381) runs on the GPU for all instances (riccati_kernel.hip.h).
int tiny_batch_setup_hetero(TinyBatch** out, const double* Adyn, const double* Bdyn, const double* fdyn, const double* Qdiag,
                            const double* Rdiag, const double* rho, int nx, int nu, int N, int batch, int device, int verbose) {
    if (!out || !Adyn || !Bdyn || !Qdiag || !Rdiag || !rho) return TINY_ERR_NULL;
    if (nx <= 0 || nu <= 0 || nx + nu > 32 || nu > 16) return TINY_ERR_UNSUPPORTED;
    // instance 0 builds the ordinary handle (records, shared tables for bounds / cones / masks, settings)
    int rc = tiny_batch_setup(out, Adyn, Bdyn, fdyn, Qdiag, Rdiag, rho[0], nx, nu, N, batch, device, verbose);
    if (rc) return rc;
    TinyBatch* b = *out;
    // the kernel that will read the per-instance tables decides their layout: the one-row kernel where it holds the shape, else the
    // tile kernel's per-instance form (wide and long shapes: round 5); neither: unsupported
    if (!has_regs(b) && !(b->tile && !b->no_jit)) { tiny_batch_destroy(b); *out = nullptr; return TINY_ERR_UNSUPPORTED; }
    const bool tile_layout = !has_regs(b);
    const int tab_cols = tile_layout ? 32 : 16, tab_lw = tile_layout ? 16 * std::max(1, b->tile->W) : 16;
    static_assert(het_tab_doubles(16, 16) == (int)HET_TAB_DOUBLES && het_tab_doubles(32, 16) == TileTab<1>::BOUNDS && het_tab_doubles(32, 32) == TileTab<2>::BOUNDS, "per-instance table layouts");
    auto bail = [&](int code) { tiny_batch_destroy(b); *out = nullptr; return code; };
    const size_t xx = (size_t)nx * nx, xu = (size_t)nx * nu, uu = (size_t)nu * nu, B = batch;
    struct { double** p; size_t n; } bufs[] = {{&b->d_hA, B * xx}, {&b->d_hB, B * xu}, {&b->d_hf, B * nx}, {&b->d_hQw, B * nx},
        {&b->d_hRw, B * nu}, {&b->d_hrho, B}, {&b->d_hK, B * xu}, {&b->d_hP, B * xx}, {&b->d_hQuu, B * uu}, {&b->d_hAmBKt, B * xx},
        {&b->d_hAPf, B * nx}, {&b->d_hBPf, B * nu}, {&b->d_het_tabs, B * (size_t)het_tab_doubles(tab_cols, tab_lw)}};
    for (auto& u : bufs)
        if (hipMalloc(u.p, u.n * sizeof(double)) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMalloc(&b->d_hiters, B * sizeof(int)) != hipSuccess) return bail(TINY_ERR_HIP);
    std::vector<double> qw(B * nx), rw(B * nu), fz(B * nx, 0.0);
    for (size_t i = 0; i < B; ++i) {
        for (int k = 0; k < nx; ++k) qw[i * nx + k] = Qdiag[i * nx + k] + rho[i];     // work->Q = diag(Q) + rho (tiny_api.cpp:117)
        for (int k = 0; k < nu; ++k) rw[i * nu + k] = Rdiag[i * nu + k] + rho[i];     // work->R (tiny_api.cpp:118)
    }
    if (hipMemcpy(b->d_hA, Adyn, B * xx * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemcpy(b->d_hB, Bdyn, B * xu * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemcpy(b->d_hf, fdyn ? fdyn : fz.data(), B * nx * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemcpy(b->d_hQw, qw.data(), B * nx * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemcpy(b->d_hRw, rw.data(), B * nu * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemcpy(b->d_hrho, rho, B * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(TINY_ERR_HIP);
    RiccatiArgs r;
    r.A = b->d_hA; r.B = b->d_hB; r.f = b->d_hf; r.Qw = b->d_hQw; r.Rw = b->d_hRw; r.rho = b->d_hrho;
    r.Kinf = b->d_hK; r.Pinf = b->d_hP; r.Quu_inv = b->d_hQuu; r.AmBKt = b->d_hAmBKt; r.APf = b->d_hAPf; r.BPf = b->d_hBPf;
    r.iters = b->d_hiters; r.tabs = b->d_het_tabs; r.nx = nx; r.nu = nu; r.batch = batch; r.tab_cols = tab_cols; r.tab_lw = tab_lw;
    const size_t lds = (7 * xx + 7 * xu + 3 * uu + 2 * nx + nu) * sizeof(double);
    int grid = batch < b->num_cus * 8 ? batch : b->num_cus * 8;
    if (launch_riccati(b, r, lds, grid) != TINY_OK) return bail(TINY_ERR_HIP);
    std::vector<int> its(B);
    if (hipMemcpy(its.data(), b->d_hiters, B * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return bail(TINY_ERR_HIP);
    for (size_t i = 0; i < B; ++i)
        if (its[i] < 0) { fail(b, TINY_ERR_ARG, "singular R + B'PB for instance %zu", i); return bail(TINY_ERR_ARG); }
    b->hetero = true;
    b->het_tab_cols = tab_cols; b->het_tab_lw = tab_lw;      // (the launch checks the kernel it is about to run against this layout: ADVICE r05)
    return TINY_OK;
}

// cache of ONE instance of a heterogeneous batch (names as tiny_batch_get_cache, plus "riccati_iters")
int tiny_batch_get_cache_instance(TinyBatch* b, int instance, const char* name, double* out, int capacity) {
    if (!b || !name || !out) return TINY_ERR_NULL;
    if (!b->hetero) return tiny_batch_get_cache(b, name, out, capacity);
    if (instance < 0 || instance >= b->batch) return fail(b, TINY_ERR_ARG, "instance out of range");
    const int nx = b->nx, nu = b->nu;
    const double* src = nullptr; int n = 0;
    if (!strcmp(name, "Kinf")) { src = b->d_hK; n = nu * nx; }
    else if (!strcmp(name, "Pinf")) { src = b->d_hP; n = nx * nx; }
    else if (!strcmp(name, "Quu_inv")) { src = b->d_hQuu; n = nu * nu; }
    else if (!strcmp(name, "AmBKt")) { src = b->d_hAmBKt; n = nx * nx; }
    else if (!strcmp(name, "APf")) { src = b->d_hAPf; n = nx; }
    else if (!strcmp(name, "BPf")) { src = b->d_hBPf; n = nu; }
    else if (!strcmp(name, "Q")) { src = b->d_hQw; n = nx; }
    else if (!strcmp(name, "R")) { src = b->d_hRw; n = nu; }
    else if (!strcmp(name, "riccati_iters")) {
        int it = 0;
        HIP_TRY(b, hipMemcpy(&it, b->d_hiters + instance, sizeof(int), hipMemcpyDeviceToHost));
        if (capacity >= 1) out[0] = it;
        return 1;
    } else return fail(b, TINY_ERR_ARG, "unknown cache member %s", name);
    if (capacity >= n) HIP_TRY(b, hipMemcpy(out, src + (size_t)instance * n, n * sizeof(double), hipMemcpyDeviceToHost));
    return n;
}

int tiny_batch_destroy(TinyBatch* b) {
    if (!b) return TINY_ERR_NULL;
    hipSetDevice(b->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    void* bufs[] = {b->d_kpi_slab, b->d_x0,
                    b->d_stage, b->d_status, b->d_resid, b->d_stats, b->d_tab, b->d_dbg_qr, b->d_dbg_pd, b->d_accum,
                    b->d_iter_log, b->d_u0_log, b->d_lslack, b->d_ldual, b->d_tlslack, b->d_tldual, b->d_gtab, b->d_traj,
                    b->d_traj_offsets, b->d_hA, b->d_hB, b->d_hf, b->d_hQw, b->d_hRw, b->d_hrho, b->d_hK, b->d_hP, b->d_hQuu,
                    b->d_hAmBKt, b->d_hAPf, b->d_hBPf, b->d_het_tabs, b->d_hiters, b->d_ttab, b->d_repack_index, b->d_repack_count,
                    b->d_wire, b->d_arho, b->d_aK, b->d_aP, b->d_aC1, b->d_aC2, b->d_atab, b->d_work_counter, b->d_perm, b->d_rg_bins, b->d_ls, b->d_pf_counter, b->d_vz_scratch};
    for (void* p : bufs)
        if (p) hipFree(p);
    if (b->h_wire) hipHostFree(b->h_wire);
    if (b->d_hist) hipFree(b->d_hist);
    if (b->h_hist) hipHostFree(b->h_hist);
    if (b->hist_ev) hipEventDestroy(b->hist_ev);
    if (b->h_ls) hipHostFree(b->h_ls);
    if (b->ls_ev) hipEventDestroy(b->ls_ev);
    if (b->rg_fork) hipEventDestroy(b->rg_fork);
    if (b->rg_join) hipEventDestroy(b->rg_join);
    if (b->stream2) { hipStreamSynchronize(b->stream2); hipStreamDestroy(b->stream2); }
    if (b->auto_ev0) hipEventDestroy(b->auto_ev0);
    if (b->auto_ev1) hipEventDestroy(b->auto_ev1);
    for (hipEvent_t e : b->ev_start) hipEventDestroy(e);
    for (hipEvent_t e : b->ev_stop) hipEventDestroy(e);
    if (b->own_stream && b->stream) hipStreamDestroy(b->stream);
    delete b;
    return TINY_OK;
}

const char* tiny_batch_last_error(TinyBatch* b) { return b ? b->err : "null batch"; }

int tiny_batch_set_bound_constraints(TinyBatch* b, const double* x_min, const double* x_max, const double* u_min,
                                     const double* u_max) {
    if (!b) { printf("Error in tiny_set_bound_constraints: solver is nullptr\n"); return 1; }   // tiny_api.cpp:152-155
    if (!x_min || !x_max || !u_min || !u_max) return fail(b, TINY_ERR_NULL, "null bound pointer");
    const size_t ns = (size_t)b->nx * b->N, ni = (size_t)b->nu * (b->N - 1);
    b->x_min.assign(x_min, x_min + ns); b->x_max.assign(x_max, x_max + ns);
    b->u_min.assign(u_min, u_min + ni); b->u_max.assign(u_max, u_max + ni);
    b->have_bounds = true;
    b->tab_dirty = true;
    return TINY_OK;
}

int tiny_batch_set_cone_constraints(TinyBatch* b, int nsc, const int* Acx, const int* qcx, const double* cx,
                                    int nic, const int* Acu, const int* qcu, const double* cu) {
    if (!b) { printf("Error in tiny_set_cone_constraints: solver is nullptr\n"); return 1; }     // tiny_api.cpp:179-182
    if (nsc < 0 || nic < 0) return fail(b, TINY_ERR_DIM, "negative cone count");
    if ((nsc > 0 && (!Acx || !qcx || !cx)) || (nic > 0 && (!Acu || !qcu || !cu))) return fail(b, TINY_ERR_NULL, "null cone descriptor with a positive count");
    std::vector<int> used((size_t)(b->nx + b->nu), 0);
    bool overlap_x = false, overlap_u = false;
    for (int pass = 0; pass < 2; ++pass) {
        const int n = pass ? nic : nsc;
        const int* A = pass ? Acu : Acx;
        const int* q = pass ? qcu : qcx;
        const int dim = pass ? b->nu : b->nx, off = pass ? b->nx : 0;
        for (int k = 0; k < n; ++k) {
            if (q[k] != 3) return fail(b, TINY_ERR_UNSUPPORTED, "cone dimension %d: the reference's project_soc only handles 3 (admm.cpp:53)", q[k]);
            if (A[k] < 0 || A[k] + 3 > dim) return fail(b, TINY_ERR_DIM, "cone %d out of range", k);
            // The reference projects the cones of a column one after the other whether they share rows or not
            // (admm.cpp:111-135).  The register-resident kernels give every row to at most one cone; families with
            // overlapping cones are served by the coverage kernel, which walks the cones in the reference's order.
            for (int c3 = 0; c3 < 3; ++c3)
                if (used[off + A[k] + c3]++) (pass ? overlap_u : overlap_x) = true;
        }
    }
    b->cones_overlap_x = overlap_x; b->cones_overlap_u = overlap_u;
    b->Acx.assign(Acx, Acx + nsc); b->qcx.assign(qcx, qcx + nsc); b->cx.assign(cx, cx + nsc);
    b->Acu.assign(Acu, Acu + nic); b->qcu.assign(qcu, qcu + nic); b->cu.assign(cu, cu + nic);
    b->tab_dirty = true;
    return TINY_OK;
}

int tiny_batch_update_settings(TinyBatch* b, double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                               int en_state_bound, int en_input_bound, int en_state_soc, int en_input_soc,
                               int en_state_linear, int en_input_linear, int en_tv_state_linear,
                               int en_tv_input_linear) {
    if (!b) { printf("Error in tiny_update_settings: settings is nullptr\n"); return 1; }        // tiny_api.cpp:393-396
    b->set.abs_pri_tol = abs_pri_tol; b->set.abs_dua_tol = abs_dua_tol; b->set.max_iter = max_iter;
    b->set.check_termination = check_termination; b->set.en_state_bound = en_state_bound;
    b->set.en_input_bound = en_input_bound; b->set.en_state_soc = en_state_soc; b->set.en_input_soc = en_input_soc;
    b->set.en_state_linear = en_state_linear; b->set.en_input_linear = en_input_linear;
    b->set.en_tv_state_linear = en_tv_state_linear; b->set.en_tv_input_linear = en_tv_input_linear;
    b->tab_dirty = true;
    return TINY_OK;
}

// column-major (n x cols) -> row-major [k][cols]
static void rows_of(const double* colmajor, int n, int cols, std::vector<double>* out) {
    out->assign((size_t)n * cols, 0.0);
    for (int k = 0; k < n; ++k)
        for (int c = 0; c < cols; ++c) (*out)[(size_t)k * cols + c] = colmajor[(size_t)c * n + k];
}

int tiny_batch_set_linear_constraints(TinyBatch* b, int n_state, const double* Alin_x, const double* blin_x, int n_input,
                                      const double* Alin_u, const double* blin_u) {
    if (!b) { printf("Error in tiny_set_linear_constraints: solver is nullptr\n"); return 1; }   // tiny_api.cpp:213-216
    if (n_state < 0 || n_input < 0) return fail(b, TINY_ERR_DIM, "negative constraint count");
    if ((n_state > 0 && (!Alin_x || !blin_x)) || (n_input > 0 && (!Alin_u || !blin_u))) return fail(b, TINY_ERR_NULL, "null constraint table with a positive count");
    b->nsl = n_state; b->nil = n_input;
    rows_of(Alin_x, n_state, b->nx, &b->Alin_x); b->blin_x.assign(blin_x, blin_x + n_state);
    rows_of(Alin_u, n_input, b->nu, &b->Alin_u); b->blin_u.assign(blin_u, blin_u + n_input);
    b->tab_dirty = true;
    return TINY_OK;
}

int tiny_batch_set_tv_linear_constraints(TinyBatch* b, int n_state, const double* tv_Alin_x, const double* tv_blin_x,
                                         int n_input, const double* tv_Alin_u, const double* tv_blin_u) {
    if (!b) { printf("Error in tiny_set_linear_constraints: solver is nullptr\n"); return 1; }   // tiny_api.cpp:256-259
    if (n_state < 0 || n_input < 0) return fail(b, TINY_ERR_DIM, "negative constraint count");
    if ((n_state > 0 && (!tv_Alin_x || !tv_blin_x)) || (n_input > 0 && (!tv_Alin_u || !tv_blin_u))) return fail(b, TINY_ERR_NULL, "null constraint table with a positive count");
    const int N = b->N;
    b->ntsl = n_state; b->ntil = n_input;
    // tv_Alin_x is (n_state*N) x nx column-major with row n_state*i + k = constraint k at knot i (admm.cpp:189):
    // row-major it is already [knot][k][nx]; tv_blin_x is n_state x N column-major = [knot][k]
    rows_of(tv_Alin_x, n_state * N, b->nx, &b->tvA_x); b->tvb_x.assign(tv_blin_x, tv_blin_x + (size_t)n_state * N);
    rows_of(tv_Alin_u, n_input * (N - 1), b->nu, &b->tvA_u); b->tvb_u.assign(tv_blin_u, tv_blin_u + (size_t)n_input * (N - 1));
    b->tab_dirty = true;
    return TINY_OK;
}

int tiny_batch_get_cache(TinyBatch* b, const char* name, double* out, int capacity) {
    if (!b || !name) return TINY_ERR_NULL;
    const std::vector<double>* v = nullptr;
    if (!strcmp(name, "Kinf")) v = &b->cache.Kinf.a;
    else if (!strcmp(name, "Pinf")) v = &b->cache.Pinf.a;
    else if (!strcmp(name, "Quu_inv")) v = &b->cache.Quu_inv.a;
    else if (!strcmp(name, "AmBKt")) v = &b->cache.AmBKt.a;
    else if (!strcmp(name, "APf")) v = &b->cache.APf.a;
    else if (!strcmp(name, "BPf")) v = &b->cache.BPf.a;
    else if (!strcmp(name, "Q")) v = &b->Qw;
    else if (!strcmp(name, "R")) v = &b->Rw;
    else return fail(b, TINY_ERR_ARG, "unknown cache member %s", name);
    if (out && capacity >= (int)v->size()) memcpy(out, v->data(), v->size() * sizeof(double));
    return (int)v->size();
}

// Overwrite one cache member of the family (names as tiny_batch_get_cache, plus "rho": one double): for callers that bring
// their own cache instead of tiny_setup's Riccati recursion -- generated code (tiny_codegen) restores a frozen TinyCache with it.
int tiny_batch_set_cache(TinyBatch* b, const char* name, const double* src) {
    if (!b || !name || !src) return TINY_ERR_NULL;
    if (b->hetero) return fail(b, TINY_ERR_UNSUPPORTED, "heterogeneous batches keep per-instance caches (tiny_batch_setup_hetero)");
    std::vector<double>* v = nullptr;
    if (!strcmp(name, "rho")) { b->cache.rho = src[0]; }
    else if (!strcmp(name, "Kinf")) v = &b->cache.Kinf.a;
    else if (!strcmp(name, "Pinf")) v = &b->cache.Pinf.a;
    else if (!strcmp(name, "Quu_inv")) v = &b->cache.Quu_inv.a;
    else if (!strcmp(name, "AmBKt")) v = &b->cache.AmBKt.a;
    else if (!strcmp(name, "APf")) v = &b->cache.APf.a;
    else if (!strcmp(name, "BPf")) v = &b->cache.BPf.a;
    else if (!strcmp(name, "Q")) v = &b->Qw;
    else if (!strcmp(name, "R")) v = &b->Rw;
    else return fail(b, TINY_ERR_ARG, "unknown cache member %s", name);
    if (v) memcpy(v->data(), src, v->size() * sizeof(double));
    b->tab_dirty = true;
    if (b->d_arho) { HIP_TRY(b, hipSetDevice(b->device)); if (int rc = adaptive_fresh_state(b)) return rc; }   // adaptive state restarts from it
    return TINY_OK;
}

static int prepare_field(TinyBatch* b, TinyField field, double** kpi, int* rows, int* row_off, int* cols) {
    if (field >= TINY_F_Q && field <= TINY_F_D) {            // inputs of the single-phase entry points only (tiny_batch_phase)
        if (int rc = ensure_debug_buffers(b)) return rc;
    }
    if (field >= TINY_F_VLNEW && field < TINY_F_COUNT) {     // linear-constraint records are allocated on first use
        double** arr[] = {&b->d_lslack, &b->d_lslack, &b->d_ldual, &b->d_ldual, &b->d_tlslack, &b->d_tlslack, &b->d_tldual, &b->d_tldual};
        if (int rc = ensure_kpi(b, arr[field - TINY_F_VLNEW])) return rc;
    }
    if (field_geometry(b, field, kpi, rows, row_off, cols)) return fail(b, TINY_ERR_ARG, "bad field %d", (int)field);
    if (!*kpi) return fail(b, TINY_ERR_ARG, "field %d has no record yet (constraint family never enabled)", (int)field);
    return TINY_OK;
}

}  // extern "C"
namespace tinympc_amd {
// fields[i] <-> d_buf + offsets[i] (device memory, host layout [batch][cols][rows]) in ONE launch; TINY_F_X0 is the nx-vector x0;
// with_status (download only): status int4 [batch] and the residuals [batch][4] follow at off_status / off_resid
int xfer_fields(TinyBatch* b, const TinyField* fields, const size_t* offsets, int n, double* d_buf, bool to_device,
                bool with_status, size_t off_status, size_t off_resid) {
    if (n + (with_status ? 2 : 0) > 40) return fail(b, TINY_ERR_ARG, "too many fields in one transfer");
    HIP_TRY(b, hipSetDevice(b->device));
    XferTable t;
    t.n = 0;
    for (int i = 0; i < n; ++i) {
        XferEntry& e = t.e[t.n++];
        e.off = (long)offsets[i];
        if (fields[i] == TINY_F_X0) { e.kpi = nullptr; e.raw = b->d_x0; e.count = (long)b->batch * b->nx; e.rows = e.row_off = e.cols = 0; continue; }
        if (int rc = prepare_field(b, fields[i], &e.kpi, &e.rows, &e.row_off, &e.cols)) return rc;
        e.raw = nullptr; e.count = 0;
        if (to_device && fields[i] != TINY_F_XREF && fields[i] != TINY_F_UREF) b->records_zero = false;
        if (to_device && fields[i] == TINY_F_XREF) b->xref_shared = false;
        if (to_device && fields[i] == TINY_F_UREF) b->uref_shared = false;
    }
    if (with_status && !to_device) {
        XferEntry& s = t.e[t.n++];
        s.kpi = nullptr; s.raw = reinterpret_cast<double*>(b->d_status); s.count = (long)b->batch * 2; s.off = (long)off_status; s.rows = s.row_off = s.cols = 0;
        XferEntry& r = t.e[t.n++];
        r.kpi = nullptr; r.raw = b->d_resid; r.count = (long)b->batch * 4; r.off = (long)off_resid; r.rows = r.row_off = r.cols = 0;
    }
    const size_t per = (size_t)b->batch * b->N * b->nx;
    const int gx = (int)std::min<size_t>(512, (per + 255) / 256);
    hipLaunchKernelGGL(xfer_fields_kernel, dim3(gx > 0 ? gx : 1, t.n), dim3(256), 0, b->stream, t, d_buf, b->batch, b->N, b->nx + b->nu, to_device ? 1 : 0);
    HIP_TRY(b, hipGetLastError());
    return TINY_OK;
}
}  // namespace tinympc_amd
extern "C" {

int tiny_batch_set(TinyBatch* b, TinyField field, const double* src, int flags) {
    if (!b || !src) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    const bool dev = flags & TINY_DEVICE, bc = flags & TINY_BROADCAST;
    const int nx = b->nx;
    if (field == TINY_F_X0) {
        const size_t n = (size_t)(bc ? 1 : b->batch) * nx;
        if (!bc) {
            HIP_TRY(b, hipMemcpyAsync(b->d_x0, src, n * sizeof(double), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, b->stream));
        } else {
            const double* s = src;
            if (!dev) { HIP_TRY(b, hipMemcpyAsync(b->d_stage, src, n * sizeof(double), hipMemcpyHostToDevice, b->stream)); s = b->d_stage; }
            hipLaunchKernelGGL(broadcast_rows_kernel, dim3(1024), dim3(256), 0, b->stream, b->d_x0, s, b->batch, nx);
            HIP_TRY(b, hipGetLastError());
        }
        if (!dev) HIP_TRY(b, hipStreamSynchronize(b->stream));
        return TINY_OK;
    }
    if (field >= TINY_F_Q && field <= TINY_F_D) {            // inputs of the single-phase entry points only (tiny_batch_phase)
        if (int rc = ensure_debug_buffers(b)) return rc;
    }
    if (field >= TINY_F_VLNEW && field < TINY_F_COUNT) {     // linear-constraint records are allocated on first use
        double** arr[] = {&b->d_lslack, &b->d_lslack, &b->d_ldual, &b->d_ldual, &b->d_tlslack, &b->d_tlslack, &b->d_tldual, &b->d_tldual};
        if (int rc = ensure_kpi(b, arr[field - TINY_F_VLNEW])) return rc;
    }
    double* kpi; int rows, row_off, cols;
    if (field_geometry(b, field, &kpi, &rows, &row_off, &cols)) return fail(b, TINY_ERR_ARG, "bad field %d", (int)field);
    if (field != TINY_F_XREF && field != TINY_F_UREF) b->records_zero = false;     // a warm-start record was written by the caller
    if (field == TINY_F_XREF) b->xref_shared = bc;          // one reference for every instance: launches read one record
    if (field == TINY_F_UREF) b->uref_shared = bc;
    const size_t n = (size_t)(bc ? 1 : b->batch) * rows * cols;
    const double* s = src;
    if (!dev) { HIP_TRY(b, hipMemcpyAsync(b->d_stage, src, n * sizeof(double), hipMemcpyHostToDevice, b->stream)); s = b->d_stage; }
    hipLaunchKernelGGL(pack_kpi_kernel, dim3(2048), dim3(256), 0, b->stream, kpi, s, b->batch, b->N, b->nx + b->nu, rows,
                       row_off, cols, bc ? 1 : 0);
    HIP_TRY(b, hipGetLastError());
    if (!dev) HIP_TRY(b, hipStreamSynchronize(b->stream));   // the staging buffer is reused by the next call
    return TINY_OK;
}

int tiny_batch_get(TinyBatch* b, TinyField field, double* dst, int flags) {
    if (!b || !dst) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    const bool dev = flags & TINY_DEVICE;
    if (field == TINY_F_X0) {
        HIP_TRY(b, hipMemcpyAsync(dst, b->d_x0, (size_t)b->batch * b->nx * sizeof(double), dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, b->stream));
        if (!dev) HIP_TRY(b, hipStreamSynchronize(b->stream));
        return TINY_OK;
    }
    if (field >= TINY_F_Q && field <= TINY_F_D && !b->debug && !b->d_dbg_qr)
        return fail(b, TINY_ERR_ARG, "q/r/p/d need set_option(\"debug\", 1) before the solve");
    double* kpi; int rows, row_off, cols;
    if (field_geometry(b, field, &kpi, &rows, &row_off, &cols)) return fail(b, TINY_ERR_ARG, "bad field %d", (int)field);
    if (!kpi) return fail(b, TINY_ERR_ARG, "field %d has no record yet (constraint family never enabled)", (int)field);
    const size_t n = (size_t)b->batch * rows * cols;
    double* d = dev ? dst : b->d_stage;
    hipLaunchKernelGGL(unpack_kpi_kernel, dim3(2048), dim3(256), 0, b->stream, kpi, d, b->batch, b->N, b->nx + b->nu, rows,
                       row_off, cols);
    HIP_TRY(b, hipGetLastError());
    if (!dev) {
        HIP_TRY(b, hipMemcpyAsync(dst, b->d_stage, n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(b, hipStreamSynchronize(b->stream));
    }
    return TINY_OK;
}

int tiny_batch_reset(TinyBatch* b) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    const size_t kpi_bytes = (size_t)b->batch * b->N * (b->nx + b->nu) * sizeof(double);
    double* z[] = {b->d_prim, b->d_slack, b->d_dual, b->d_slack_prev, b->d_cslack, b->d_cdual, b->d_lslack, b->d_ldual,
                   b->d_tlslack, b->d_tldual};
    for (double* p : z)
        if (p) HIP_TRY(b, hipMemsetAsync(p, 0, kpi_bytes, b->stream));
    HIP_TRY(b, hipMemsetAsync(b->d_accum, 0, (size_t)b->batch * sizeof(uint2), b->stream));
    if (b->d_arho && !b->astate_fresh)               // adaptive rho: every instance's cache back to the one tiny_setup computed
        if (int rc = adaptive_fresh_state(b)) return rc;
    b->records_zero = true;                           // the next one-row launch need not READ what it knows to be zero (launch_solve)
    b->status_valid = false;                          // (d_status: the last episode's counts say nothing about the next one's first step)
    return TINY_OK;
}

// == settings->adaptive_rho, adaptive_rho_min / _max / _enable_clipping (types.hpp:75-79)
int tiny_batch_set_adaptive_rho(TinyBatch* b, int enable, double rho_min, double rho_max, int enable_clipping) {
    if (!b) return TINY_ERR_NULL;
    b->adaptive = enable != 0; b->adaptive_min = rho_min; b->adaptive_max = rho_max; b->adaptive_clip = enable_clipping != 0;
    return TINY_OK;
}

// == cache->dKinf_drho (nu x nx), dPinf_drho (nx x nx), dC1_drho (nu x nu), dC2_drho (nx x nx), column-major, shared by every
// instance (what tiny_initialize_sensitivity_matrices fills, tiny_api.cpp:479-540); dC1 / dC2 may be NULL (taken as zero)
int tiny_batch_set_sensitivity(TinyBatch* b, const double* dKinf, const double* dPinf, const double* dC1, const double* dC2) {
    if (!b || !dKinf || !dPinf) return TINY_ERR_NULL;
    const size_t nx = b->nx, nu = b->nu;
    b->dKinf.assign(dKinf, dKinf + nu * nx);
    b->dPinf.assign(dPinf, dPinf + nx * nx);
    if (dC1) b->dC1.assign(dC1, dC1 + nu * nu); else b->dC1.clear();
    if (dC2) b->dC2.assign(dC2, dC2 + nx * nx); else b->dC2.clear();
    b->atab_dirty = true;
    return TINY_OK;
}

// Per-instance cache state of an adaptive batch, host arrays with a leading batch axis (column-major matrices): which =
// "rho" [batch], "Kinf" [batch][nu*nx], "Pinf" [batch][nx*nx], "C1" [batch][nu*nu], "C2" [batch][nx*nx].
static int cache_state_array(TinyBatch* b, const char* which, double** arr, size_t* per) {
    if (!strcmp(which, "rho")) { *arr = b->d_arho; *per = 1; }
    else if (!strcmp(which, "Kinf")) { *arr = b->d_aK; *per = (size_t)b->nu * b->nx; }
    else if (!strcmp(which, "Pinf")) { *arr = b->d_aP; *per = (size_t)b->nx * b->nx; }
    else if (!strcmp(which, "C1")) { *arr = b->d_aC1; *per = (size_t)b->nu * b->nu; }
    else if (!strcmp(which, "C2")) { *arr = b->d_aC2; *per = (size_t)b->nx * b->nx; }
    else return fail(b, TINY_ERR_ARG, "unknown cache state %s", which);
    return TINY_OK;
}
int tiny_batch_set_cache_state(TinyBatch* b, const char* which, const double* src) {
    if (!b || !which || !src) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    if (!b->adaptive) return fail(b, TINY_ERR_ARG, "the cache is per-instance state only with adaptive rho on (tiny_batch_set_adaptive_rho)");
    if (!b->d_arho) { if (int rc = ensure_adaptive(b, false)) return rc; }
    double* arr; size_t per;
    if (int rc = cache_state_array(b, which, &arr, &per)) return rc;
    HIP_TRY(b, hipMemcpyAsync(arr, src, (size_t)b->batch * per * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    b->astate_fresh = false;
    return TINY_OK;
}
int tiny_batch_get_cache_state(TinyBatch* b, const char* which, double* dst) {
    if (!b || !which || !dst) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    if (!b->adaptive) return fail(b, TINY_ERR_ARG, "the cache is per-instance state only with adaptive rho on (tiny_batch_set_adaptive_rho)");
    if (!b->d_arho) { if (int rc = ensure_adaptive(b, false)) return rc; }
    double* arr; size_t per;
    if (int rc = cache_state_array(b, which, &arr, &per)) return rc;
    HIP_TRY(b, hipMemcpyAsync(dst, arr, (size_t)b->batch * per * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    return TINY_OK;
}

int tiny_batch_solve_async(TinyBatch* b) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    return launch_solve(b);
}

int tiny_batch_phase(TinyBatch* b, int phase) {
    if (!b) return TINY_ERR_NULL;
    if (phase < PHASE_LINEAR_COST || phase > PHASE_TERMINATION) return fail(b, TINY_ERR_ARG, "phase %d: expected 1..6", phase);
    HIP_TRY(b, hipSetDevice(b->device));
    b->records_zero = false;
    return launch_general(b, phase);
}

int tiny_batch_synchronize(TinyBatch* b) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    return TINY_OK;
}

int tiny_batch_reduce_stats(TinyBatch* b, double* host_out, void* device_out) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    double* dst = device_out ? (double*)device_out : b->d_stats;
    HIP_TRY(b, hipMemsetAsync(dst, 0, 10 * sizeof(double), b->stream));
    int blocks = (b->batch + 1023) / 1024;                // four instances per thread at most 64 blocks: 512 atomics
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(reduce_stats_kernel, dim3(blocks), dim3(256), 0, b->stream, b->d_status, b->d_resid, b->d_accum,
                       b->batch, dst);
    HIP_TRY(b, hipGetLastError());
    if (host_out) {
        HIP_TRY(b, hipMemcpyAsync(host_out, dst, 10 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(b, hipStreamSynchronize(b->stream));
    }
    return TINY_OK;
}

int tiny_batch_solve(TinyBatch* b) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    if (int rc = launch_solve(b)) return rc;
    double st[10];
    if (int rc = tiny_batch_reduce_stats(b, st, nullptr)) return rc;
    return (st[1] == (double)b->batch) ? 0 : 1;          // tiny_solve: 0 converged, 1 max_iter reached
}

int tiny_batch_get_status(TinyBatch* b, int* iter, int* solved, int* status, double* residuals) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    std::vector<int4> st(b->batch);
    HIP_TRY(b, hipMemcpyAsync(st.data(), b->d_status, (size_t)b->batch * sizeof(int4), hipMemcpyDeviceToHost, b->stream));
    if (residuals)
        HIP_TRY(b, hipMemcpyAsync(residuals, b->d_resid, (size_t)b->batch * 4 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    for (int i = 0; i < b->batch; ++i) {
        if (iter) iter[i] = st[i].x;
        if (solved) solved[i] = st[i].y;
        if (status) status[i] = st[i].z;
    }
    return TINY_OK;
}

int tiny_batch_set_option(TinyBatch* b, const char* name, long value) {
    if (!b || !name) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    if (!strcmp(name, "advance_x0")) b->advance_x0 = value != 0;
    else if (!strcmp(name, "debug")) {
        b->debug = value != 0;
        if (b->debug) { if (int rc = ensure_debug_buffers(b)) return rc; }
    } else if (!strcmp(name, "grid_waves_per_cu")) b->grid_waves_per_cu = (int)value;
    else if (!strcmp(name, "dpp_mode")) b->dpp_mode = (int)value;
    else if (!strcmp(name, "steps_per_launch")) b->steps_per_launch = (int)value;
    else if (!strcmp(name, "force_general")) { b->force_general = value != 0; b->tab_dirty = true; }
    else if (!strcmp(name, "no_jit")) { b->no_jit = value != 0; b->tab_dirty = true; if (!b->no_jit) b->variant_jit_failed = false; }
    else if (!strcmp(name, "no_tile")) { b->no_tile = value != 0; b->tab_dirty = true; }
    else if (!strcmp(name, "prefer_tile")) { b->prefer_tile = value != 0; b->tab_dirty = true; }
    else if (!strcmp(name, "tile_dyn")) b->tile_dyn_opt = (int)value;   // -1 (default): by batch size; 0: static tiles; 1: the dynamic form whenever it exists
    else if (!strcmp(name, "het_ub")) b->het_ub = value != 0;
    else if (!strcmp(name, "tile_w")) b->tile_w = (int)value;      // experiments: only entries with this W (0 = half rows, -1 = any)
    else if (!strcmp(name, "tile_lm")) b->tile_lm = (int)value;    // experiments: the tile_dims.txt entry with this LM column
    else if (!strcmp(name, "tile_r")) b->tile_r = (int)value;      // experiments: the tile_dims.txt entry with this R (0: the first that fits)
    else if (!strcmp(name, "step_log")) b->step_log = value != 0;
    else if (!strcmp(name, "reset_duals")) b->reset_duals = value != 0;
    else if (!strcmp(name, "store_primal")) { if (value < 0 || value > 2) return fail(b, TINY_ERR_ARG, "store_primal: 0, 1 or 2"); b->store_primal = (int)value; }
    else if (!strcmp(name, "share_ref")) b->share_ref = value != 0;
    else if (!strcmp(name, "half_rows")) b->half_rows = (int)value;
    else if (!strcmp(name, "one_shot_fast")) b->one_shot_fast = value != 0;
    else if (!strcmp(name, "repack_tail")) { if (value < -1 || value > 1) return fail(b, TINY_ERR_ARG, "repack_tail: -1 (by rule), 0 (follow-up stages), 1 (the tile kernel's dynamic form)"); b->repack_tail = (int)value; }
    else if (!strcmp(name, "plan")) { b->plan_opt = value != 0 ? 1 : 0; if (!b->plan_opt) b->plan_tried = true; }
    else if (!strcmp(name, "prefetch")) { if (value < -1 || value > 1) return fail(b, TINY_ERR_ARG, "prefetch: -1 (by rule), 0 (never), 1 (wherever the form exists)"); b->prefetch = (int)value; }
    else if (!strcmp(name, "prefetch_static")) { if (value < -1 || value > 100) return fail(b, TINY_ERR_ARG, "prefetch_static: percent, 0 ... 100, or -1 (by rule: 75 warm, 50 cold)"); b->prefetch_static = (int)value; }
    else if (!strcmp(name, "prefetch_waves")) { if (value < 0) return fail(b, TINY_ERR_ARG, "prefetch_waves >= 0"); b->prefetch_waves = (int)value; }
    else if (!strcmp(name, "launch_order")) { if (value < 0 || value > 2) return fail(b, TINY_ERR_ARG, "launch_order: 0 (ascending), 1 (alternating), 2 (descending)"); b->launch_order = (int)value; }
    else if (!strcmp(name, "step_regroup")) { if (value < -1) return fail(b, TINY_ERR_ARG, "step_regroup: K > 0 (stretches of K MPC steps), 0 (never) or -1 (automatic)"); b->step_regroup = (int)value; b->regroup_verdict = 0; b->regroup_since = 0; b->ls_pending = false; }
    else if (!strcmp(name, "step_regroup_streams")) { if (value < 1 || value > 2) return fail(b, TINY_ERR_ARG, "step_regroup_streams: 1 or 2"); b->regroup_streams = (int)value; }
    else if (!strcmp(name, "auto_cold")) b->auto_cold = value != 0;       // 0: always read the warm-start records, also right after a reset
    else if (!strcmp(name, "uniform_bounds")) b->use_ub = value != 0;
    else if (!strcmp(name, "one_shot")) { if (value < 0 || value > 2) return fail(b, TINY_ERR_ARG, "one_shot: 0, 1 or 2"); b->one_shot = (int)value; }
    else if (!strcmp(name, "repack_after")) { if (value < -1) return fail(b, TINY_ERR_ARG, "repack_after: K > 0, 0 (never) or -1 (automatic)"); b->repack_after = (int)value; b->auto_cap = 0; b->hist_copy.clear(); b->hist_pending = false; b->auto_verdict = 0; b->growth_verdict = 0; b->auto_plain_rate = b->auto_split_rate = 0.0; b->auto_probes = 0; if (value < 0 && b->plan_opt) { b->plan_tried = false; b->plan_shipped = false; } }    // (automatic again: the shipped plan is looked up afresh)
    else if (!strcmp(name, "repack_waves_per_cu")) b->repack_waves_per_cu = (int)std::max(1L, value);
    else if (!strcmp(name, "repack_growth")) { b->repack_growth = (int)value; b->growth_verdict = 0; }
    else if (!strcmp(name, "repack_sort")) { if (value < -1 || value > 1) return fail(b, TINY_ERR_ARG, "repack_sort: 1, 0 or -1 (automatic)"); b->repack_sort = (int)value; }
    else if (!strcmp(name, "repack_dynamic")) b->repack_dynamic = value != 0;   // follow-up stages: tiles off a counter (1) or a fixed grid stride (0)
    else if (!strcmp(name, "traj_step")) b->traj_step = value;
    else if (!strcmp(name, "timing")) {
        HIP_TRY(b, hipStreamSynchronize(b->stream));
        while ((long)b->ev_start.size() < value) {
            hipEvent_t s, e;
            HIP_TRY(b, hipEventCreate(&s));
            HIP_TRY(b, hipEventCreate(&e));
            b->ev_start.push_back(s); b->ev_stop.push_back(e);
        }
        b->timing_n = 0;
        b->timing_left = (int)value;
    } else return fail(b, TINY_ERR_ARG, "unknown option %s", name);
    return TINY_OK;
}

int tiny_batch_set_stream(TinyBatch* b, void* hip_stream) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    if (b->own_stream && b->stream) hipStreamDestroy(b->stream);
    b->stream = (hipStream_t)hip_stream;
    b->own_stream = false;
    return TINY_OK;
}

int tiny_batch_get_timing(TinyBatch* b, float* ms, int capacity) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    for (int i = 0; i < b->timing_n && i < capacity; ++i)
        HIP_TRY(b, hipEventElapsedTime(&ms[i], b->ev_start[i], b->ev_stop[i]));
    return b->timing_n;
}

int tiny_batch_set_reference_trajectory(TinyBatch* b, const double* xref_points, int n_points, const int* offsets, int flags) {
    if (!b) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    if (b->d_traj) { hipFree(b->d_traj); b->d_traj = nullptr; }
    if (b->d_traj_offsets) { hipFree(b->d_traj_offsets); b->d_traj_offsets = nullptr; }
    b->traj_points = 0; b->traj_step = 0;
    if (!xref_points || n_points <= 0) return TINY_OK;                 // back to per-instance Xref records
    const bool dev = flags & TINY_DEVICE;
    const size_t bytes = (size_t)n_points * b->nx * sizeof(double);
    HIP_TRY(b, hipMalloc(&b->d_traj, bytes));
    HIP_TRY(b, hipMemcpy(b->d_traj, xref_points, bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    if (offsets) {
        HIP_TRY(b, hipMalloc(&b->d_traj_offsets, (size_t)b->batch * sizeof(int)));
        HIP_TRY(b, hipMemcpy(b->d_traj_offsets, offsets, (size_t)b->batch * sizeof(int), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    }
    b->traj_points = n_points;
    return TINY_OK;
}

int tiny_batch_get_step_log(TinyBatch* b, int* iters, double* u0, int steps) {
    if (!b) return TINY_ERR_NULL;
    if (steps > b->log_steps || !b->d_iter_log) return fail(b, TINY_ERR_ARG, "no step log recorded (set_option step_log/steps_per_launch)");
    HIP_TRY(b, hipSetDevice(b->device));
    if (iters) HIP_TRY(b, hipMemcpyAsync(iters, b->d_iter_log, (size_t)steps * b->batch * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    if (u0) HIP_TRY(b, hipMemcpyAsync(u0, b->d_u0_log, (size_t)steps * b->batch * b->nu * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    return TINY_OK;
}

// which kernel the next tiny_batch_solve would run: 0 one-row register kernel, 1 tile kernel, 2 coverage kernel
long tiny_jit_compile(const char* instantiation, int* from_disk, char* msg, int msg_len) {
    std::string why;
    const long n = jit_compile_only(instantiation, from_disk, &why);
    if (msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", why.c_str());
    return n > 0 ? n : (long)TINY_ERR_HIP;
}
long tiny_jit_prebuild(const char* instantiation, const char* dir, char* msg, int msg_len) {
    std::string why;
    const long n = jit_prebuild(instantiation, dir, &why);
    if (msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", why.c_str());
    return n >= 0 ? n : (long)TINY_ERR_HIP;
}
int tiny_jit_used(char* out, int out_len) {
    std::string names;
    const int n = jit_used_names(&names);
    if (out && out_len > 0) snprintf(out, (size_t)out_len, "%s", names.c_str());
    return n;
}

// The cost model of the automatic split solve by itself (host arithmetic, no GPU): hist[i] = instances whose solve takes i
// iterations (1024 bins).  Returns the proposed K (0: a plain launch is predicted within 5 %), *ratio = predicted time of the
// best split / plain launch.  What launch_solve consults with the histogram of the previous solve.
int tiny_predict_split(const unsigned* hist, int nx, int nu, int N, int max_iter, int check_termination, int num_cus, double* ratio) {
    if (!hist || nx <= 0 || nu <= 0 || N < 2 || max_iter <= 0) return 0;
    return choose_split_for(nx, nu, N, false, max_iter, std::max(1, check_termination), 0, num_cus > 0 ? num_cus : 256, hist, ratio);
}

// The schedule "step_regroup" cuts a fused launch of `steps` MPC steps into (host arithmetic, no GPU): k > 0 = stretches of k steps,
// k <= 0 = the automatic length; `known` = the instances' last iteration counts are known at the start (else ONE step goes first: it
// is what tells them apart); half = 1: the schedule of the second half of the batch under "step_regroup_streams" = 2 (half a stretch
// out of step).  Writes the stretch lengths to out[0 .. capacity) and returns their number.
int tiny_step_regroup_plan(int steps, int k, int known, int half, int* out, int capacity) {
    if (steps < 1) return 0;
    const int K = k > 0 ? k : regroup_auto_k(steps);
    std::vector<int> plan;
    int lead = 0;
    if (!known && steps > 1) { plan.push_back(1); lead = 1; }
    if (K < steps)
        for (const int n : regroup_stretches(steps - lead, K, (half && regroup_two_streams_apply(steps, lead, K)) ? (K + 1) / 2 : 0)) plan.push_back(n);
    else plan.assign(1, steps);
    for (int i = 0; i < (int)plan.size() && i < capacity; ++i) out[i] = plan[i];
    return (int)plan.size();
}

// read-back of derived state: "auto_split_k" (the K the automatic split picked from the last histogram, 0 = plain launch),
// "auto_split_permille" (its predicted time in 1/1000 of the plain launch's), "repack_after"
long tiny_batch_get_option(TinyBatch* b, const char* name) {
    if (!b || !name) return TINY_ERR_NULL;
    if (!strcmp(name, "auto_split_k")) {
        if (b->hist_pending && hipEventSynchronize(b->hist_ev) == hipSuccess) {      // (a diagnostic may wait; a solve never does)
            b->hist_pending = false;
            b->probe_was_tile = b->probe_was_growth = false;   // (the probe's clock reading is dropped with it: the next probe starts clean)
            if (b->auto_verdict == 0) {                        // (a decided batch keeps its K AND the stage schedule the clock chose)
                b->auto_cap = choose_split(b, b->h_hist, &b->auto_gain);
                b->auto_cap_max_iter = b->set.max_iter;
                b->hist_copy.assign(b->h_hist, b->h_hist + TinyBatch::HIST_BINS);
            }
        }
        return b->auto_cap;
    }
    if (!strcmp(name, "auto_split_growth_verdict")) return b->growth_verdict;
    if (!strcmp(name, "auto_split_growth")) return b->auto_growth;               // the stage schedule that goes with auto_split_k: K, g K, g^2 K, ...
    if (!strcmp(name, "auto_split_permille")) return (long)(b->auto_gain * 1000.0 + 0.5);
    if (!strcmp(name, "auto_split_verdict")) return b->auto_verdict;
    if (!strcmp(name, "tile_alt_verdict")) return b->tile_verdict;              // 1: the one-row shape runs on the tile kernel's dynamic form (the clock said so), -1: it does not
    if (!strcmp(name, "last_prefetch_grid")) return b->last_pf_grid;         // ... its persistent grid (waves) and its tile buffer (bytes of dynamic LDS)
    if (!strcmp(name, "last_prefetch_lds")) return (long)b->last_pf_lds;
    if (!strcmp(name, "last_tail_tile")) return b->last_tail_tile ? 1 : 0;     // the last split solve ran its tail on the tile kernel's dynamic form
    if (!strcmp(name, "plan_shipped")) return b->plan_shipped ? 1 : 0;        // a plan of data/plans.txt was imported at the first solve
    if (!strcmp(name, "last_prefetch")) return b->last_prefetch ? 1 : 0;     // the last one-row launch (a split solve: its first stage) took the PREFETCH form
    if (!strcmp(name, "last_half_rows")) return b->last_half ? 1 : 0;       // the last one-row launch took the HALF form (two instances per DPP row)
    if (!strcmp(name, "last_tile_form")) return b->last_tile_form;          // W * 1e6 + R * 1e3 + LM of the tile_dims.txt entry the last tile launch took (-1: run-time instantiated)
    if (!strcmp(name, "last_tile_dyn")) return b->last_tile_dyn ? 1 : 0;      // the last tile-kernel launch took the dynamic slot form
    if (!strcmp(name, "auto_split_measured_permille")) return (b->auto_plain_rate > 0.0 && b->auto_split_rate > 0.0) ? (long)(1000.0 * b->auto_split_rate / b->auto_plain_rate + 0.5) : 0;
    if (!strcmp(name, "repack_after")) return b->repack_after;
    if (!strcmp(name, "repack_sorted_stages")) return b->last_sorted_stages;       // follow-up stages of the last split solve that took a sorted list
    if (!strcmp(name, "step_regroup")) return b->step_regroup;
    if (!strcmp(name, "step_regroup_stretches")) return b->last_regroup_stretches;   // launches the last fused solve was cut into (1: not cut)
    if (!strcmp(name, "step_regroup_verdict") || !strcmp(name, "lockstep_permille")) {
        if (b->ls_pending && hipEventSynchronize(b->ls_ev) == hipSuccess) read_lockstep_estimate(b);      // (a diagnostic may wait; a solve never does)
        return name[0] == 's' ? (long)b->regroup_verdict : (long)(b->lockstep_ratio * 1000.0 + 0.5);
    }
    return fail(b, TINY_ERR_ARG, "unknown option %s", name);
}

// ---- the settled launch form of a batch as plain data -----------------------------------------------------------
// What the clock-checked dispatch has learnt about a batch -- plain or split solve and its K / stage schedule, the histogram the
// schedule came from (repack_sort's stage predictions), the tile kernel's dynamic form for a one-row shape, stretches of MPC steps
// for a fused launch -- leaves the process as one POD and enters another one: the importing handle takes the settled form on its
// FIRST solve instead of spending six solves on probes, and two boxes given the same plan launch the same way.
int tiny_batch_get_plan(TinyBatch* b, TinyBatchPlan* out) {
    if (!b || !out) return TINY_ERR_NULL;
    HIP_TRY(b, hipSetDevice(b->device));
    // (a diagnostic may wait for what the last solve left behind; a solve never does)
    if (b->hist_pending && hipEventSynchronize(b->hist_ev) == hipSuccess) learn_from_probe(b, b->set.max_iter, true);
    if (b->ls_pending && hipEventSynchronize(b->ls_ev) == hipSuccess) read_lockstep_estimate(b);
    memset(out, 0, sizeof(*out));
    out->magic = TINY_PLAN_MAGIC; out->version = TINY_PLAN_VERSION; out->bytes = (int)sizeof(TinyBatchPlan);
    out->nx = b->nx; out->nu = b->nu; out->N = b->N; out->batch = b->batch;
    out->max_iter = b->set.max_iter; out->check_termination = b->set.check_termination;
    out->auto_verdict = b->auto_verdict; out->auto_cap = b->auto_cap; out->auto_cap_max_iter = b->auto_cap_max_iter;
    out->auto_growth = b->auto_growth; out->growth_verdict = b->growth_verdict; out->auto_probes = b->auto_probes;
    out->auto_plain_rate = b->auto_plain_rate; out->auto_split_rate = b->auto_split_rate; out->auto_gain = b->auto_gain;
    out->tile_verdict = b->tile_verdict; out->tile_rate = b->tile_rate;
    out->regroup_verdict = b->regroup_verdict; out->lockstep_ratio = b->lockstep_ratio;
    out->hist_valid = b->hist_copy.size() == (size_t)TinyBatch::HIST_BINS ? 1 : 0;
    if (out->hist_valid) memcpy(out->hist, b->hist_copy.data(), sizeof(out->hist));
    // open questions: plain-or-split undecided (while a split is on the table), the other stage schedule of a kept split untried
    out->open_questions = ((b->auto_verdict == 0 && !(b->auto_plain_rate > 0.0 && b->auto_cap == 0)) ? 1 : 0) +
                          ((b->auto_verdict == 1 && b->growth_verdict == 0 && b->repack_growth < 2) ? 1 : 0);
    return TINY_OK;
}

int tiny_batch_set_plan(TinyBatch* b, const TinyBatchPlan* in) {
    if (!b || !in) return TINY_ERR_NULL;
    if (in->magic != TINY_PLAN_MAGIC || in->version != TINY_PLAN_VERSION || in->bytes != (int)sizeof(TinyBatchPlan))
        return fail(b, TINY_ERR_ARG, "not a TinyBatchPlan of this library version");
    if (in->nx != b->nx || in->nu != b->nu || in->N != b->N)
        return fail(b, TINY_ERR_DIM, "the plan was made for (nx,nu,N)=(%d,%d,%d), this batch is (%d,%d,%d)", in->nx, in->nu, in->N, b->nx, b->nu, b->N);
    // every field is checked (a plan is a POD read from a file): verdicts in {-1, 0, 1}, counts non-negative, clock
    // readings finite and non-negative.  A plan is advice only -- results never depend on it -- but a corrupt one must not pin a launch form
    auto verdict = [](int v) { return v >= -1 && v <= 1; };
    auto rate = [](double v) { return v >= 0.0 && v < 1e300; };
    if (!verdict(in->auto_verdict) || !verdict(in->tile_verdict) || !verdict(in->regroup_verdict) || !verdict(in->growth_verdict) ||
        in->auto_cap < 0 || in->auto_cap >= TinyBatch::HIST_BINS || (in->auto_growth != 2 && in->auto_growth != 4) || in->auto_cap_max_iter < 0 ||
        in->auto_probes < 0 || in->batch <= 0 || !rate(in->auto_plain_rate) || !rate(in->auto_split_rate) || !rate(in->auto_gain) || !rate(in->tile_rate) ||
        !rate(in->lockstep_ratio) || (in->hist_valid != 0 && in->hist_valid != 1))
        return fail(b, TINY_ERR_ARG, "TinyBatchPlan: field out of range");
    HIP_TRY(b, hipSetDevice(b->device));
    // (a plan is advice about launch forms, never about results: every form is bit-identical.  It applies to solves with the
    // max_iter it was made for -- auto_cap_max_iter -- exactly as a plan learnt in this process would)
    // A verdict whose supporting clock reading is missing, or that was made under other settings than the handle's, counts as OPEN
    const bool same_settings = in->max_iter == b->set.max_iter && in->check_termination == b->set.check_termination;
    const bool far_batch = in->batch > 4L * b->batch || b->batch > 4L * in->batch;       // (the plan of a batch of another order of magnitude)
    b->auto_verdict = (same_settings && !far_batch && in->auto_plain_rate > 0.0 && (in->auto_verdict != 1 || in->auto_split_rate > 0.0)) ? in->auto_verdict : 0;
    b->auto_cap = in->auto_cap; b->auto_cap_max_iter = in->auto_cap_max_iter;
    b->auto_growth = in->auto_growth; b->growth_verdict = b->auto_verdict == 1 ? in->growth_verdict : 0; b->auto_probes = std::max(2, in->auto_probes);
    b->auto_plain_rate = in->auto_plain_rate; b->auto_split_rate = in->auto_split_rate; b->auto_gain = in->auto_gain;
    b->tile_verdict = (same_settings && !far_batch && (in->tile_verdict != 1 || in->tile_rate > 0.0)) ? in->tile_verdict : 0; b->tile_rate = in->tile_rate;
    b->regroup_verdict = far_batch ? 0 : in->regroup_verdict; b->lockstep_ratio = in->lockstep_ratio;
    if (in->hist_valid) b->hist_copy.assign(in->hist, in->hist + TinyBatch::HIST_BINS); else b->hist_copy.clear();
    b->auto_since = b->tile_since = b->regroup_since = 0;
    b->plan_tried = true;                            // (an imported plan stands: the shipped one is not looked up behind it)
    b->hist_pending = false; b->ls_pending = false; b->probe_was_tile = false; b->probe_was_growth = false; b->auto_last_cap = 0;
    // what the first launch of an imported form would otherwise allocate inside its solve call
    if (b->auto_verdict == 1 && b->auto_cap > 0) {
        if (int rc = ensure_repack_buffers(b)) return rc;
        if (b->repack_sort != 0) { if (int rc = ensure_regroup_buffers(b, false)) return rc; }
    }
    if (b->regroup_verdict == 1) { if (int rc = ensure_regroup_buffers(b, true)) return rc; }
    return TINY_OK;
}

}  // extern "C"
namespace tinympc_amd {
// ---- shipped plans: tinympc_amd/data/plans.txt next to the library (or TINYMPC_AMD_PLANS=<file>; "0": none) ---------------------------
// one plan per line, written by tools/make_plans.py from tiny_batch_get_plan of settled handles:
//   plan | plan_soc <mask>   nx nu N batch max_iter check_termination auto_verdict auto_cap auto_cap_max_iter auto_growth growth_verdict auto_probes
//        tile_verdict regroup_verdict auto_plain_rate auto_split_rate auto_gain tile_rate lockstep_ratio nhist i:count ...
// a line may begin with `plan_soc <mask>` instead of `plan`: the entry then only serves handles whose ACTIVE cone families are that mask
// (bit 0 inputs, bit 1 states; BASELINE config 4: the three cone settings of one shape settle on different launch forms -- stretches of
// MPC steps pay with the thrust cone alone, not with the state cone, whose rows already iterate alike)
static std::vector<int> g_plan_soc;                    // parallel to shipped_plans(): -1 = any
static const std::vector<TinyBatchPlan>& shipped_plans() {
    static std::vector<TinyBatchPlan> plans;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string path;
        if (const char* e = getenv("TINYMPC_AMD_PLANS")) path = e;
        else {
            Dl_info info;
            if (dladdr(reinterpret_cast<const void*>(&tiny_batch_setup), &info) && info.dli_fname) {
                path = info.dli_fname;
                const size_t cut = path.find_last_of('/');
                path = (cut == std::string::npos ? std::string(".") : path.substr(0, cut)) + "/data/plans.txt";
            }
        }
        if (path.empty() || path == "0") return;
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return;
        char word[16];
        while (fscanf(f, "%15s", word) == 1) {
            int soc_mask = -1;
            if (!strcmp(word, "plan_soc")) { if (fscanf(f, "%d", &soc_mask) != 1) break; }
            else if (strcmp(word, "plan") != 0) { int c; while ((c = fgetc(f)) != EOF && c != '\n') {} continue; }      // comments, unknown lines
            TinyBatchPlan p;
            memset(&p, 0, sizeof(p));
            p.magic = TINY_PLAN_MAGIC; p.version = TINY_PLAN_VERSION; p.bytes = (int)sizeof(TinyBatchPlan);
            int nh = 0;
            if (fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %lf %lf %lf %lf %lf %d", &p.nx, &p.nu, &p.N, &p.batch, &p.max_iter, &p.check_termination,
                       &p.auto_verdict, &p.auto_cap, &p.auto_cap_max_iter, &p.auto_growth, &p.growth_verdict, &p.auto_probes, &p.tile_verdict, &p.regroup_verdict,
                       &p.auto_plain_rate, &p.auto_split_rate, &p.auto_gain, &p.tile_rate, &p.lockstep_ratio, &nh) != 20) break;
            bool ok = nh >= 0 && nh <= 1024;
            for (int i = 0; i < nh && ok; ++i) {
                int bin = 0; unsigned cnt = 0;
                ok = fscanf(f, "%d:%u", &bin, &cnt) == 2 && bin >= 0 && bin < 1024;
                if (ok) p.hist[bin] = cnt;
            }
            if (!ok) break;
            p.hist_valid = nh > 0 ? 1 : 0;
            plans.push_back(p);
            g_plan_soc.push_back(soc_mask);
        }
        fclose(f);
    });
    return plans;
}
bool apply_shipped_plan(TinyBatch* b) {
    const TinyBatchPlan* best = nullptr;
    double best_d = 1e300;
    const std::vector<TinyBatchPlan>& all = shipped_plans();
    const int soc_now = ((b->set.en_input_soc && !b->Acu.empty()) ? 1 : 0) | ((b->set.en_state_soc && !b->Acx.empty()) ? 2 : 0);
    for (size_t i = 0; i < all.size(); ++i) {
        const TinyBatchPlan& p = all[i];
        if (g_plan_soc[i] >= 0 && g_plan_soc[i] != soc_now) continue;
        if (p.nx != b->nx || p.nu != b->nu || p.N != b->N || p.max_iter != b->set.max_iter || p.check_termination != b->set.check_termination) continue;
        if (p.batch > 2L * b->batch || b->batch > 2L * p.batch) continue;                 // the batch bucket: within a factor of two
        const double d = std::fabs(std::log((double)p.batch / (double)b->batch));
        if (d < best_d) { best_d = d; best = &p; }
    }
    if (!best) return false;
    if (tiny_batch_set_plan(b, best) != TINY_OK) { b->err[0] = 0; return false; }
    b->plan_shipped = true;
    return true;
}
}  // namespace tinympc_amd
extern "C" {

int tiny_batch_kernel_path(TinyBatch* b) {
    if (!b) return TINY_ERR_NULL;
    if (use_tile(b)) return b->tile_is_jit ? 4 : 1;
    if (use_general(b)) return 2;
    return b->kernel ? 0 : 3;                          // 3: the one-row kernel, instantiated at run time
}

long tiny_batch_algorithmic_bytes(TinyBatch* b, int cold) {
    if (!b) return 0;
    const long S = (long)b->nx * b->N + (long)b->nu * (b->N - 1);
    return 8 * (b->nx + (cold == 2 ? 3 : (cold ? 2 : 8)) * S) + 44;      // cold 1: bytes_cold; 2: one_shot = 1 (x|u and vnew|znew out)
}

}  // extern "C"
