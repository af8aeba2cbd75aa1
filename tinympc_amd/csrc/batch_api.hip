// batch_api.hip -- host side of the batched, device-resident C ABI (include/tinympc_amd.h, part A).
//
// Mirrors the reference operator interface for the hot path (src/tinympc/tiny_api.cpp:21-147
// setup, :149-208 constraint setters, :388-411 settings, :443-477 x0/xref/uref, :384-386 solve)
// with a leading batch axis; the ADMM iteration itself is admm_kernel.hip.h.  No CPU fallback.
#define TINYMPC_GENERAL_KERNEL_IMPL
#include "batch_impl.hpp"

#include <cstdarg>
#include <cstdio>
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

// ---- kernel registry: one translation unit per (nx, nu, N) of kernel_dims.txt (generated by the Makefile)
#include "_gen/registry.inc"
static const int g_nkernels = (int)(sizeof(g_kernels) / sizeof(g_kernels[0]));

static const int g_ntiles = (int)(sizeof(g_tiles) / sizeof(g_tiles[0]));
static const TileEntry* find_tile(int nx, int nu, int N) {
    for (int i = 0; i < g_ntiles; ++i)
        if (g_tiles[i]->nx == nx && g_tiles[i]->nu == nu && g_tiles[i]->N == N) return g_tiles[i];
    return nullptr;
}
// tile_dims.txt may list a shape with several R (rows along the horizon), the preferred one first; every entry of a shape has
// the same W (table layout).  The plain launch takes the first entry that HAS the needed form (an R = 1 entry of a long horizon has
// no box-table-in-LDS form: one wave's LDS would not hold it) -- or the entry with R == want_r (option "tile_r", experiments).
// R of the cone / half-space variants of a compiled-in shape (all arrays in registers, trajectory in LDS): the LAST entry of the
// shape in tile_dims.txt, i.e. the one with the most rows along the horizon
static int variant_tile_r(const TinyBatch* b) {
    int r = b->tile ? b->tile->R : 1;
    for (int i = 0; i < g_ntiles; ++i)
        if (g_tiles[i]->nx == b->nx && g_tiles[i]->nu == b->nu && g_tiles[i]->N == b->N) r = g_tiles[i]->R;
    return r;
}
// the box is the same at every knot (what build_tables / build_tile_tables_w find out from the tables they build; here from the
// host copies of the bounds, for decisions that are taken before a table exists)
static bool box_is_uniform(const TinyBatch* b) {
    if (b->N < 2) return false;
    if (!b->have_bounds) return true;
    const int nx = b->nx, nu = b->nu, N = b->N;
    if (b->set.en_state_bound)
        for (int i = 1; i < N; ++i)
            for (int j = 0; j < nx; ++j)
                if (b->x_min[(size_t)i * nx + j] != b->x_min[j] || b->x_max[(size_t)i * nx + j] != b->x_max[j]) return false;
    if (b->set.en_input_bound)
        for (int i = 1; i < N - 1; ++i)
            for (int a = 0; a < nu; ++a)
                if (b->u_min[(size_t)i * nu + a] != b->u_min[a] || b->u_max[(size_t)i * nu + a] != b->u_max[a]) return false;
    return true;
}
// static LDS of a cone / half-space variant of the tile kernel at (w, r): bound tables (the UB form keeps two slots), the trajectory,
// the cone slack's three planes (rows of the families that are on), two planes per half-space set (all rows) and the sets' tables
static long variant_tile_lds_bytes(const TinyBatch* b, int w, int r, int socm, int lv, int km, bool ub) {
    const long LW = 16L * w, N = b->N, ipw = 4 / (w * r), nz = b->nx + b->nu;
    const long cr = ((socm & 2) ? b->nx : 0) + ((socm & 1) ? b->nu : 0), csr = (cr + 1) | 1, csl = (nz + 1) | 1;
    long d = 2L * (ub ? 2 : N) * LW + (N / r) * 64;
    if (socm) d += (ipw * N + 1) * 3 * csr;
    if (lv & 1) d += 3L * km * LW + 2L * ipw * N * csl;
    if (lv & 2) d += 3L * N * km * LW + 2L * ipw * N * csl;
    return 8L * d;
}
// R of a cone / half-space variant of a compiled-in shape: its slacks live in LDS planes, so five L-long arrays are all a lane
// holds -- the smallest R whose arrays fit the register file of one wave per SIMD and whose planes, tables and trajectory fit the
// wave's static LDS (a horizon split over R rows leaves all but one of them idle in the sweeps); 0: none does
static int budget_variant_tile_r(const TinyBatch* b, int socm, int lv, int km, bool ub) {
    const int w = std::max(1, b->tile->W), nz = b->nx + b->nu;
    for (int r = 1; r <= 4 / w; r *= 2) {
        if (b->N % r || b->N / r < 2) continue;
        if (2 * (5 * (b->N / r) + 2 * nz) + 44 + 24 <= 512 && variant_tile_lds_bytes(b, w, r, socm, lv, km, ub) <= TILE_LDS_STATIC_LIMIT) return r;
    }
    return 0;
}
static const TileEntry* pick_tile_entry(const TinyBatch* b, bool ub) {
    const TileEntry* first_ok = nullptr;
    for (int i = 0; i < g_ntiles; ++i) {
        const TileEntry* t = g_tiles[i];
        if (t->nx != b->nx || t->nu != b->nu || t->N != b->N) continue;
        if (!(ub ? (t->kub != nullptr || t->k != nullptr) : (t->k != nullptr))) continue;
        if (b->tile_w >= 0 && t->W != b->tile_w) continue;                  // option "tile_w" (experiments): 0 = half rows
        if (b->tile_r > 0 && t->R == b->tile_r && (b->tile_lm < 0 || t->lm == b->tile_lm)) return t;
        if (b->tile_r == 0 && b->tile_lm >= 0 && t->lm == b->tile_lm) return t;
        if (!first_ok) first_ok = t;
    }
    return first_ok;
}
static const KernelEntry* find_kernel(int nx, int nu, int N) {
    for (int i = 0; i < g_nkernels; ++i)
        if (g_kernels[i]->nx == nx && g_kernels[i]->nu == nu && g_kernels[i]->N == N) return g_kernels[i];
    return nullptr;
}

int fail(TinyBatch* b, int code, const char* fmt, ...) {
    if (b) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(b->err, sizeof(b->err), fmt, ap);
        va_end(ap);
    }
    return code;
}
#define HIP_TRY(b, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (void)hipGetLastError();   /* reported here: do not let it resurface in a later, unrelated call */ \
            return fail(b, TINY_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_));               \
        }                                                                                         \
    } while (0)

// Half-spaces per knot and family the register-resident LIN variants are built for: 4 (compiled in), 8 / 16 / 32
// (instantiated at run time; the tables must still fit the 64 KiB of static LDS), 0 = more than that: coverage kernel.
static int lin_kmax(const TinyBatch* b) {
    int m = 0;
    if (b->set.en_state_linear) m = std::max(m, b->nsl);
    if (b->set.en_input_linear) m = std::max(m, b->nil);
    if (b->set.en_tv_state_linear) m = std::max(m, b->ntsl);
    if (b->set.en_tv_input_linear) m = std::max(m, b->ntil);
    if (m <= LIN_KMAX) return LIN_KMAX;
    // more than the compiled-in variants hold: the table stride doubles (run-time instantiated KMAX = 8, 16, 32) while
    // the tables still fit the wave's static LDS
    const bool tv = b->set.en_tv_state_linear || b->set.en_tv_input_linear;
    const bool st = b->set.en_state_linear || b->set.en_input_linear;
    for (int km = 2 * LIN_KMAX; km <= LIN_KMAX_BIG; km *= 2) {
        if (m > km) continue;
        const long lds = 8L * ((tv ? 3L * b->N * km * 16 : 1) + (st ? 3L * km * 16 : 1) + b->nx * 16 + 2L * b->N * 16);
        return lds <= 63 * 1024 ? km : 0;
    }
    return 0;
}

// Lane tables for the kernel (layout in admm_kernel.hip.h).  Rebuilt whenever cache, bounds, cones or
// the enable switches change.
static void build_tables(TinyBatch* b) {
    const int nx = b->nx, nu = b->nu, N = b->N;
    const Cache& c = b->cache;
    std::vector<double>& t = b->h_tab;
    const int km = std::max(lin_kmax(b), (int)LIN_KMAX);            // stride of the half-space tables
    t.assign(tab_doubles(N, km), 0.0);
    // Quu_inv * B'  and  Quu_inv * BPf : the input rows of the fused backward step
    //   d_i = Quu_inv (B' p_{i+1} + r_i + BPf)            (admm.cpp:17)
    Mat QBt = c.Quu_inv * transpose(b->B);
    Mat QBPf = c.Quu_inv * c.BPf;
    for (int j = 0; j < nx; ++j) {                 // state lanes
        for (int k = 0; k < nx; ++k) {
            t[TAB_MB + k * 16 + j] = c.AmBKt(j, k);           // p_i += AmBKt p_{i+1}     (admm.cpp:18)
            t[TAB_MF1 + k * 16 + j] = b->A(j, k);             // x_{i+1} = A x_i ...      (admm.cpp:30)
            t[TAB_PT + k * 16 + j] = c.Pinf(k, j);            // (Xref' Pinf)[j]          (admm.cpp:292)
        }
        for (int m = 0; m < nu; ++m) {
            t[TAB_MB + (nx + m) * 16 + j] = -c.Kinf(m, j);    // - Kinf' r_i
            t[TAB_MF2 + (nx + m) * 16 + j] = b->B(j, m);      // + B u_i
        }
        t[TAB_VEC + VEC_CB * 16 + j] = c.APf(j, 0);
        t[TAB_VEC + VEC_CF * 16 + j] = b->f(j, 0);
        t[TAB_VEC + VEC_QR * 16 + j] = b->Qw[j];
        t[TAB_VEC + VEC_SMASK * 16 + j] = 1.0;
    }
    for (int a = 0; a < nu; ++a) {                 // input lanes
        const int j = nx + a;
        for (int k = 0; k < nx; ++k) {
            t[TAB_MB + k * 16 + j] = QBt(a, k);
            t[TAB_MF1 + k * 16 + j] = -c.Kinf(a, k);          // u_i = -Kinf x_i - d_i     (admm.cpp:29)
        }
        for (int m = 0; m < nu; ++m) t[TAB_MB + (nx + m) * 16 + j] = c.Quu_inv(a, m);
        t[TAB_VEC + VEC_CB * 16 + j] = QBPf(a, 0);
        t[TAB_VEC + VEC_QR * 16 + j] = b->Rw[a];
        t[TAB_VEC + VEC_NIM * 16 + j] = -1.0;
    }
    // cones (admm.cpp:102-135): lane flags
    for (int j = 0; j < 16; ++j) t[TAB_VEC + VEC_CONE_BASE * 16 + j] = -1.0;
    const bool s_on = b->set.en_state_soc && !b->Acx.empty();
    const bool i_on = b->set.en_input_soc && !b->Acu.empty();
    for (int j = 0; j < nx; ++j) t[TAB_VEC + VEC_SOCFLAG * 16 + j] = s_on ? 1.0 : 0.0;
    for (int a = 0; a < nu; ++a) t[TAB_VEC + VEC_SOCFLAG * 16 + nx + a] = i_on ? 1.0 : 0.0;
    if (b->set.en_state_soc)
        for (size_t k = 0; k < b->Acx.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) {
                t[TAB_VEC + VEC_CONE_BASE * 16 + b->Acx[k] + c3] = b->Acx[k];
                t[TAB_VEC + VEC_CONE_MU * 16 + b->Acx[k] + c3] = b->cx[k];
            }
    if (b->set.en_input_soc)
        for (size_t k = 0; k < b->Acu.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) {
                t[TAB_VEC + VEC_CONE_BASE * 16 + nx + b->Acu[k] + c3] = nx + b->Acu[k];
                t[TAB_VEC + VEC_CONE_MU * 16 + nx + b->Acu[k] + c3] = b->cu[k];
            }
    // linear-constraint slacks exist for a whole family as soon as its switch is on (admm.cpp:138-145, 176-183)
    for (int j = 0; j < nx; ++j) {
        t[TAB_VEC + VEC_LINFLAG * 16 + j] = b->set.en_state_linear ? 1.0 : 0.0;
        t[TAB_VEC + VEC_TLINFLAG * 16 + j] = b->set.en_tv_state_linear ? 1.0 : 0.0;
    }
    for (int a = 0; a < nu; ++a) {
        t[TAB_VEC + VEC_LINFLAG * 16 + nx + a] = b->set.en_input_linear ? 1.0 : 0.0;
        t[TAB_VEC + VEC_TLINFLAG * 16 + nx + a] = b->set.en_tv_input_linear ? 1.0 : 0.0;
    }
    // bounds (admm.cpp:91-98): a disabled or never-set box is (-inf, +inf)
    const double inf = std::numeric_limits<double>::infinity();
    {   // half-space tables of the LIN kernel variants: [k][16] coefficient, offset, squared norm
        auto fill = [&](double* blk, const double* Arow, int n, double bk, int lane0, bool enabled, int k) {
            double nn = 0.0;
            for (int c = 0; c < n; ++c) nn += Arow[c] * Arow[c];
            for (int c = 0; c < n; ++c) {
                blk[k * 16 + lane0 + c] = enabled ? Arow[c] : 0.0;
                blk[km * 16 + k * 16 + lane0 + c] = enabled ? bk : inf;
                blk[2 * km * 16 + k * 16 + lane0 + c] = enabled ? nn : 1.0;
            }
        };
        double* ls = &t[tab_lin_offset(N)];
        for (int k = 0; k < km; ++k)
            for (int j = 0; j < 16; ++j) { ls[km * 16 + k * 16 + j] = inf; ls[2 * km * 16 + k * 16 + j] = 1.0; }
        for (int k = 0; k < b->nsl && k < km; ++k) fill(ls, &b->Alin_x[(size_t)k * nx], nx, b->blin_x[k], 0, b->set.en_state_linear, k);
        for (int k = 0; k < b->nil && k < km; ++k) fill(ls, &b->Alin_u[(size_t)k * nu], nu, b->blin_u[k], nx, b->set.en_input_linear, k);
        double* lt = &t[tab_tlin_offset(N, km)];
        for (int s = 0; s < N; ++s) {
            double* blk = lt + (size_t)s * 3 * km * 16;
            for (int k = 0; k < km; ++k)
                for (int j = 0; j < 16; ++j) { blk[km * 16 + k * 16 + j] = inf; blk[2 * km * 16 + k * 16 + j] = 1.0; }
            for (int k = 0; k < b->ntsl && k < km; ++k)            // state lanes: slot s = knot s
                fill(blk, &b->tvA_x[((size_t)s * b->ntsl + k) * nx], nx, b->tvb_x[(size_t)s * b->ntsl + k], 0, b->set.en_tv_state_linear, k);
            if (s >= 1)                                                  // input lanes: slot s = knot s-1
                for (int k = 0; k < b->ntil && k < km; ++k)
                    fill(blk, &b->tvA_u[((size_t)(s - 1) * b->ntil + k) * nu], nu, b->tvb_u[(size_t)(s - 1) * b->ntil + k], nx, b->set.en_tv_input_linear, k);
        }
    }
    double* lo = &t[TAB_BOUNDS];
    double* hi = &t[TAB_BOUNDS + N * 16];
    for (int e = 0; e < N * 16; ++e) { lo[e] = -inf; hi[e] = inf; }
    if (b->set.en_state_bound && b->have_bounds)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < nx; ++j) {
                lo[i * 16 + j] = b->x_min[(size_t)i * nx + j];
                hi[i * 16 + j] = b->x_max[(size_t)i * nx + j];
            }
    if (b->set.en_input_bound && b->have_bounds)     // input lanes keep knot i in slot i+1 (admm_kernel.hip.h)
        for (int i = 0; i < N - 1; ++i)
            for (int a = 0; a < nu; ++a) {
                lo[(i + 1) * 16 + nx + a] = b->u_min[(size_t)i * nu + a];
                hi[(i + 1) * 16 + nx + a] = b->u_max[(size_t)i * nu + a];
            }
    // knot-invariant box? (slot 0 of the input lanes is the dummy slot and keeps (-inf, +inf): slots 1.. must agree; state lanes: 0..)
    bool uniform = N >= 2;
    for (int j = 0; j < nx + nu && uniform; ++j)
        for (int i = (j < nx ? 0 : 1); i < N && uniform; ++i)
            uniform = lo[i * 16 + j] == lo[16 + j] && hi[i * 16 + j] == hi[16 + j];
    b->bounds_uniform = uniform;
}

// Tables of the tile kernel (tile_kernel.hip.h): matrices [column k][LW = 16 W lanes], vectors [LW], bounds [N][LW]
template <int W>
static void build_tile_tables_w(TinyBatch* b) {
    using T = TileTab<W>;
    const int nx = b->nx, nu = b->nu, N = b->N, LW = T::LW;
    const Cache& c = b->cache;
    std::vector<double>& t = b->h_ttab;
    const int km = std::max(lin_kmax(b), (int)LIN_KMAX);
    t.assign(T::doubles(N, km), 0.0);
    Mat QBt = c.Quu_inv * transpose(b->B);
    Mat QBPf = c.Quu_inv * c.BPf;
    for (int j = 0; j < nx; ++j) {
        for (int k = 0; k < nx; ++k) {
            t[T::MB + k * LW + j] = c.AmBKt(j, k);
            t[T::MF1 + k * LW + j] = b->A(j, k);
            t[T::PT + k * LW + j] = c.Pinf(k, j);
        }
        for (int m = 0; m < nu; ++m) {
            t[T::MB + (nx + m) * LW + j] = -c.Kinf(m, j);
            t[T::MF2 + (nx + m) * LW + j] = b->B(j, m);
        }
        t[T::VEC + VEC_CB * LW + j] = c.APf(j, 0);
        t[T::VEC + VEC_CF * LW + j] = b->f(j, 0);
        t[T::VEC + VEC_QR * LW + j] = b->Qw[j];
        t[T::VEC + VEC_SMASK * LW + j] = 1.0;
    }
    for (int a = 0; a < nu; ++a) {
        const int j = nx + a;
        for (int k = 0; k < nx; ++k) { t[T::MB + k * LW + j] = QBt(a, k); t[T::MF1 + k * LW + j] = -c.Kinf(a, k); }
        for (int m = 0; m < nu; ++m) t[T::MB + (nx + m) * LW + j] = c.Quu_inv(a, m);
        t[T::VEC + VEC_CB * LW + j] = QBPf(a, 0);
        t[T::VEC + VEC_QR * LW + j] = b->Rw[a];
        t[T::VEC + VEC_NIM * LW + j] = -1.0;
    }
    {   // half-spaces (admm.cpp:137-211) of the LIN variants: [k][LW] coefficient, offset, squared norm (as build_tables)
        const double inf_ = std::numeric_limits<double>::infinity();
        for (int j = 0; j < nx; ++j) { t[T::VEC + VEC_LINFLAG * LW + j] = b->set.en_state_linear ? 1.0 : 0.0; t[T::VEC + VEC_TLINFLAG * LW + j] = b->set.en_tv_state_linear ? 1.0 : 0.0; }
        for (int a = 0; a < nu; ++a) { t[T::VEC + VEC_LINFLAG * LW + nx + a] = b->set.en_input_linear ? 1.0 : 0.0; t[T::VEC + VEC_TLINFLAG * LW + nx + a] = b->set.en_tv_input_linear ? 1.0 : 0.0; }
        auto fill = [&](double* blk, const double* Arow, int n, double bk, int lane0, bool enabled, int k) {
            double nn = 0.0;
            for (int c = 0; c < n; ++c) nn += Arow[c] * Arow[c];
            for (int c = 0; c < n; ++c) {
                blk[k * LW + lane0 + c] = enabled ? Arow[c] : 0.0;
                blk[km * LW + k * LW + lane0 + c] = enabled ? bk : inf_;
                blk[2 * km * LW + k * LW + lane0 + c] = enabled ? nn : 1.0;
            }
        };
        auto blank = [&](double* blk) {
            for (int k = 0; k < km; ++k)
                for (int j = 0; j < LW; ++j) { blk[km * LW + k * LW + j] = inf_; blk[2 * km * LW + k * LW + j] = 1.0; }
        };
        double* ls = &t[T::lin_offset(N)];
        blank(ls);
        for (int k = 0; k < b->nsl && k < km; ++k) fill(ls, &b->Alin_x[(size_t)k * nx], nx, b->blin_x[k], 0, b->set.en_state_linear, k);
        for (int k = 0; k < b->nil && k < km; ++k) fill(ls, &b->Alin_u[(size_t)k * nu], nu, b->blin_u[k], nx, b->set.en_input_linear, k);
        for (int sl = 0; sl < N; ++sl) {
            double* blk = &t[T::tlin_offset(N, km)] + (size_t)sl * 3 * km * LW;
            blank(blk);
            for (int k = 0; k < b->ntsl && k < km; ++k)                  // state rows: slot = knot
                fill(blk, &b->tvA_x[((size_t)sl * b->ntsl + k) * nx], nx, b->tvb_x[(size_t)sl * b->ntsl + k], 0, b->set.en_tv_state_linear, k);
            if (sl >= 1)                                                 // input rows: slot = knot + 1
                for (int k = 0; k < b->ntil && k < km; ++k)
                    fill(blk, &b->tvA_u[((size_t)(sl - 1) * b->ntil + k) * nu], nu, b->tvb_u[(size_t)(sl - 1) * b->ntil + k], nx, b->set.en_tv_input_linear, k);
        }
    }
    // cones (admm.cpp:102-135): per-row flags of the SOC variant
    for (int j = 0; j < LW; ++j) t[T::VEC + VEC_CONE_BASE * LW + j] = -1.0;
    const bool s_on = b->set.en_state_soc && !b->Acx.empty(), i_on = b->set.en_input_soc && !b->Acu.empty();
    for (int j = 0; j < nx; ++j) t[T::VEC + VEC_SOCFLAG * LW + j] = s_on ? 1.0 : 0.0;
    for (int a = 0; a < nu; ++a) t[T::VEC + VEC_SOCFLAG * LW + nx + a] = i_on ? 1.0 : 0.0;
    if (b->set.en_state_soc)
        for (size_t k = 0; k < b->Acx.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) { t[T::VEC + VEC_CONE_BASE * LW + b->Acx[k] + c3] = b->Acx[k]; t[T::VEC + VEC_CONE_MU * LW + b->Acx[k] + c3] = b->cx[k]; }
    if (b->set.en_input_soc)
        for (size_t k = 0; k < b->Acu.size(); ++k)
            for (int c3 = 0; c3 < 3; ++c3) { t[T::VEC + VEC_CONE_BASE * LW + nx + b->Acu[k] + c3] = nx + b->Acu[k]; t[T::VEC + VEC_CONE_MU * LW + nx + b->Acu[k] + c3] = b->cu[k]; }
    const double inf = std::numeric_limits<double>::infinity();
    double* lo = &t[T::BOUNDS];
    double* hi = &t[T::BOUNDS + N * LW];
    for (int e = 0; e < N * LW; ++e) { lo[e] = -inf; hi[e] = inf; }
    if (b->set.en_state_bound && b->have_bounds)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < nx; ++j) { lo[i * LW + j] = b->x_min[(size_t)i * nx + j]; hi[i * LW + j] = b->x_max[(size_t)i * nx + j]; }
    if (b->set.en_input_bound && b->have_bounds)       // input lanes keep knot i in slot i+1
        for (int i = 0; i < N - 1; ++i)
            for (int a = 0; a < nu; ++a) { lo[(i + 1) * LW + nx + a] = b->u_min[(size_t)i * nu + a]; hi[(i + 1) * LW + nx + a] = b->u_max[(size_t)i * nu + a]; }
    // knot-invariant box? (as build_tables: the input lanes' slot 0 is the dummy slot) -> the UB form of the tile kernel
    bool uniform = N >= 2;
    for (int j = 0; j < nx + nu && uniform; ++j)
        for (int i = (j < nx ? 0 : 1); i < N && uniform; ++i)
            uniform = lo[i * LW + j] == lo[LW + j] && hi[i * LW + j] == hi[LW + j];
    // (one predicate for the decision -- tile_lin_variant / use_tile budget the UB form from the host bounds -- and for the launch: the
    // table is a copy of those bounds, so the two agree by construction; should they ever not, the non-UB form is the safe one)
    b->tile_bounds_uniform = uniform && box_is_uniform(b);
}

static bool soc_active(const TinyBatch* b);
// Is there a register-resident one-row kernel for this batch: an instantiation of kernel_dims.txt, or one that hipRTC
// can make on first use (any nx + nu <= 16 whose N-long arrays fit the register file)?
static bool has_regs(const TinyBatch* b) {
    return b->kernel || (!b->no_jit && !b->jit_failed && jit_shape_fits(b->nx, b->nu, b->N, soc_active(b)));
}
static bool linear_active(const TinyBatch* b);
static int ensure_kpi(TinyBatch* b, double** p);
// LIN template value of the tile kernel's half-space variant for the current settings (run-time instantiated), 0 if the
// tables do not fit the wave's static LDS or there are more than 32 half-spaces per knot and family
static int tile_lin_variant(const TinyBatch* b) {
    if (!b->tile || !linear_active(b)) return 0;
    const int km = lin_kmax(b);
    if (km == 0) return 0;
    const int lv = ((b->set.en_state_linear || b->set.en_input_linear) ? 1 : 0) | ((b->set.en_tv_state_linear || b->set.en_tv_input_linear) ? 2 : 0);
    // (the form must exist at some R for the families that are on)
    const int socm = ((b->set.en_input_soc && !b->Acu.empty()) ? 1 : 0) | ((b->set.en_state_soc && !b->Acx.empty()) ? 2 : 0);
    const bool ub = b->use_ub && box_is_uniform(b);
    if (b->tile_is_jit) return variant_tile_lds_bytes(b, std::max(1, b->tile->W), b->tile->R, socm, lv, km, ub) <= TILE_LDS_STATIC_LIMIT ? lv : 0;
    return budget_variant_tile_r(b, socm, lv, km, ub) > 0 ? lv : 0;
}
static bool cones_overlap(const TinyBatch* b);
static int lin_variant(const TinyBatch* b);
// EXT template value of the tile kernel form this batch's launches need (tile_kernel.hip.h): bit 0 a reference-trajectory window /
// reset_duals / one_shot, bit 1 per-instance problem data; 0: the plain forms
static int tile_ext_variant(const TinyBatch* b) {
    return ((b->d_traj || b->reset_duals || b->one_shot) ? 1 : 0) | (b->hetero ? 2 : 0);
}
static bool use_tile(const TinyBatch* b) {
    if (cones_overlap(b)) return false;
    if (linear_active(b) && (tile_lin_variant(b) == 0 || b->no_jit || b->tile_soc_failed)) return false;
    // (a cone on a tile shape needs the SOC variant, which only exists through run-time instantiation)
    // a one-row shape whose half-space variant does not fit a wave's LDS (its planes grow with the horizon) takes the tile kernel's
    // (it would run its per-knot form there, at one wave per SIMD: (8,4,30) with a cone and half-spaces 72 ms against 54)
    const int lvr = linear_active(b) ? lin_variant(b) : 0;
    const bool regs_cannot = linear_active(b) && !b->force_general &&
                             (lvr == 0 || !solve_kernel_lin_planes(b->nx, b->nu, b->N, soc_active(b), lvr, lin_kmax(b), false));   // (the one-row half-space variants are instantiated without UB)
    // per-instance problem data, reference-trajectory windows, reset_duals, one_shot: the tile kernel's EXT forms (run-time instantiated
    // like its cone / half-space variants, round 5) -- for the shapes the one-row kernel does not hold; a shape it holds keeps them there
    const int ext = tile_ext_variant(b);
    if (ext && (b->no_jit || b->tile_soc_failed || has_regs(b))) return false;
    return b->tile && !((b->tile_is_jit || soc_active(b) || linear_active(b)) && (b->no_jit || b->tile_soc_failed)) && (!has_regs(b) || b->prefer_tile || regs_cannot) && !b->no_tile && !b->adaptive && !b->force_general && !b->debug;
}

// per-step iteration counts / applied controls of a fused launch (option "step_log")
static int ensure_step_logs(TinyBatch* b, int steps) {
    if (b->log_steps >= steps) return TINY_OK;
    if (b->d_iter_log) (void)hipFree(b->d_iter_log);
    if (b->d_u0_log) (void)hipFree(b->d_u0_log);
    b->d_iter_log = nullptr; b->d_u0_log = nullptr; b->log_steps = 0;
    HIP_TRY(b, hipMalloc(&b->d_iter_log, (size_t)steps * b->batch * sizeof(int)));
    HIP_TRY(b, hipMalloc(&b->d_u0_log, (size_t)steps * b->batch * b->nu * sizeof(double)));
    b->log_steps = steps;
    return TINY_OK;
}

// dry: the launch form this batch would take, over ZERO instances -- everything a first launch pays once (code object load, occupancy
// query, the work counter) without touching a record; the clock-decided dispatch does it in front of its timed probe
static int launch_tile(TinyBatch* b, bool dry = false) {
    hipFunction_t jit_fn = nullptr;
    bool jit_dyn = false;
    const bool soc = soc_active(b);
    // which families' cone slack is on (the tile kernel's SOC template value): bit 0 inputs, bit 1 states
    const int socm = ((b->set.en_input_soc && !b->Acu.empty()) ? 1 : 0) | ((b->set.en_state_soc && !b->Acx.empty()) ? 2 : 0);
    const int lv = tile_lin_variant(b);
    int vR = b->tile->R;                             // rows along the horizon of the form this launch takes
    // the tile tables follow the problem's generation, not the one-row path's dirty flag: a one-row shape whose clock-decided
    // dispatch keeps the tile form launches it from INSIDE path 0, after upload_tables() has cleared tab_dirty
    if (b->ttab_gen != b->tab_gen || b->h_ttab.empty()) {
        b->ttab_gen = b->tab_gen;
        if (b->tile->W <= 1) build_tile_tables_w<1>(b); else build_tile_tables_w<2>(b);      // (W = 0, half rows, reads the one-row tables)
        if (b->ttab_doubles < b->h_ttab.size()) {
            if (b->d_ttab) (void)hipFree(b->d_ttab);
            b->d_ttab = nullptr;
            HIP_TRY(b, hipMalloc(&b->d_ttab, b->h_ttab.size() * sizeof(double)));
            b->ttab_doubles = b->h_ttab.size();
        }
        HIP_TRY(b, hipMemcpyAsync(b->d_ttab, b->h_ttab.data(), b->h_ttab.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
    }
    const bool ub = b->tile_bounds_uniform && b->use_ub;
    const int ext = tile_ext_variant(b);
    if (b->tile_is_jit || soc || lv || ext) {        // a tile shape outside tile_dims.txt, or a cone / half-space / EXT variant: instantiate it now (jit.hpp)
        std::string why;
        // (the cone / half-space variants keep all their arrays in registers and the trajectory in LDS: their R comes from the
        // register / LDS budget of THAT form, not from the compiled-in plain form's entry)
        if (!b->tile_is_jit) {
            vR = budget_variant_tile_r(b, socm, lv, lv ? lin_kmax(b) : LIN_KMAX, ub);
            if (vR == 0) vR = variant_tile_r(b);
        }
        if (b->tile_r > 0 && (soc || lv || ext)) vR = b->tile_r;    // (option "tile_r": experiments)
        // a shape outside tile_dims.txt with plain box constraints: large batches of more than one instance per wave take the dynamic
        // slot form too (instantiated on first use like the static one; a failure falls back to the static form)
        const int jipw = 4 / (std::max(1, b->tile->W) * vR);
        jit_dyn = b->tile_is_jit && !soc && !lv && !ext && b->tile_dyn_opt != 0 && jipw >= 2 && b->grid_waves_per_cu <= 0 &&
                  (b->tile_dyn_opt > 0 || (long)((b->batch + jipw - 1) / jipw) >= 16L * b->num_cus);
        if (jit_dyn) {
            jit_fn = jit_tile_kernel(b->nx, b->nu, b->N, std::max(1, b->tile->W), vR, socm, lv, LIN_KMAX, &why, true);
            if (!jit_fn) { jit_dyn = false; why.clear(); }
        }
        if (!jit_fn) jit_fn = jit_tile_kernel(b->nx, b->nu, b->N, std::max(1, b->tile->W), vR, socm, lv, lv ? lin_kmax(b) : LIN_KMAX, &why, false, ub && (soc || lv || ext), ext);
        if (!jit_fn) {                               // the coverage kernel takes over
            if (ext) {                               // ... but not these launch forms: no other kernel runs them for this shape
                b->tile_soc_failed = true;
                return fail(b, TINY_ERR_UNSUPPORTED, "the tile kernel's form for per-instance data / reference windows / reset_duals / one_shot could not be instantiated for (%d,%d,%d): %s",
                            b->nx, b->nu, b->N, why.c_str());
            }
            if ((soc || lv) && !b->tile_is_jit) b->tile_soc_failed = true;
            else { b->tile = nullptr; b->tile_is_jit = false; }
            b->tab_dirty = true; b->redispatch = true;
            return launch_solve(b);
        }
    }
    SolveArgs a;
    a.arho = a.aK = a.aP = a.aC1 = a.aC2 = nullptr; a.atab = nullptr; a.arho_min = a.arho_max = 0.0; a.aclip = 0; a.ref_shared = 0;
    memset(&a, 0, sizeof(a));
    a.tab = b->d_ttab; a.x0 = b->d_x0; a.ref = b->d_ref; a.prim = b->d_prim; a.slack = b->d_slack; a.dual = b->d_dual;
    a.slack_prev = b->d_slack_prev; a.status = b->d_status; a.resid = b->d_resid; a.accum = b->d_accum;
    a.cslack = b->d_cslack; a.cdual = b->d_cdual;
    if (lv) {
        if (lv & 1) { if (int rc = ensure_kpi(b, &b->d_lslack)) return rc; if (int rc = ensure_kpi(b, &b->d_ldual)) return rc; }
        if (lv & 2) { if (int rc = ensure_kpi(b, &b->d_tlslack)) return rc; if (int rc = ensure_kpi(b, &b->d_tldual)) return rc; }
        a.lslack = b->d_lslack; a.ldual = b->d_ldual; a.tlslack = b->d_tlslack; a.tldual = b->d_tldual;
        a.n_lin = std::max(b->set.en_state_linear ? b->nsl : 0, b->set.en_input_linear ? b->nil : 0);
        a.n_tlin = std::max(b->set.en_tv_state_linear ? b->ntsl : 0, b->set.en_tv_input_linear ? b->ntil : 0);
    }
    const int steps = b->steps_per_launch > 1 ? b->steps_per_launch : 1;
    a.x0_next = (b->advance_x0 || steps > 1) ? b->d_x0 : nullptr;     // fused steps imply the plant step
    a.rho = b->cache.rho; a.tol_pri = b->set.abs_pri_tol; a.tol_dua = b->set.abs_dua_tol;
    a.batch = dry ? 0 : b->batch; a.max_iter = b->set.max_iter; a.check_termination = b->set.check_termination; a.steps = steps;
    a.store_mask = 31;
    if (ext) {                                       // what the EXT forms read (as launch_solve sets them for the one-row kernel)
        a.het_tabs = b->hetero ? b->d_het_tabs : nullptr;
        a.traj = b->d_traj; a.traj_offsets = b->d_traj_offsets; a.traj_points = b->traj_points; a.traj_step0 = (int)b->traj_step;
        a.reset_duals = b->reset_duals ? 1 : 0;
        a.cold = b->one_shot ? 1 : 0;
        a.store_mask = b->one_shot == 2 ? 1 : (b->one_shot == 1 ? 3 : 31);
    }
    if (steps > 1 && b->step_log) {
        if (int rc = ensure_step_logs(b, steps)) return rc;
        a.iter_log = b->d_iter_log; a.u0_log = b->d_u0_log;
    }
    const TileEntry* te = jit_fn ? nullptr : pick_tile_entry(b, ub);
    if (!jit_fn && !te) return fail(b, TINY_ERR_UNSUPPORTED, "no compiled-in tile kernel form for (%d,%d,%d)", b->nx, b->nu, b->N);
    if (te) vR = te->R;
    const int ipw = (te && te->W == 0) ? 8 / vR : 4 / (std::max(1, b->tile->W) * vR);         // instances per wave (half rows: two per DPP row)
    int grid = (b->batch + ipw - 1) / ipw;
    if (b->grid_waves_per_cu > 0) {
        const long cap = (long)b->num_cus * b->grid_waves_per_cu;
        if (cap < grid) grid = (int)cap;
    }
    const bool timed = !dry && b->timing_left > 0 && b->timing_n < (int)b->ev_start.size();
    if (timed) HIP_TRY(b, hipEventRecord(b->ev_start[b->timing_n], b->stream));

    if (jit_fn) {
        if (jit_dyn) {                              // persistent grid: what the chip holds at once
            int per_cu = 0;
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, jit_fn, 64, 0) != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); per_cu = 4; }
            if (!b->d_work_counter) HIP_TRY(b, hipMalloc(reinterpret_cast<void**>(&b->d_work_counter), sizeof(int)));
            HIP_TRY(b, hipMemsetAsync(b->d_work_counter, 0, sizeof(int), b->stream));
            a.work_counter = b->d_work_counter;
            grid = std::min(grid, per_cu * b->num_cus);
        }
        void* params[] = {&a};
        HIP_TRY(b, hipModuleLaunchKernel(jit_fn, (unsigned)grid, 1, 1, 64, 1, 1, 0, b->stream, params, nullptr));
        b->last_tile_dyn = jit_dyn;
    } else {
        // (jit_fn is null: no cone, no half-spaces.)  Dynamic form: a persistent grid -- as many waves as the chip holds at once --
        // whose slots draw instances from a device-wide counter; taken when more than one instance shares a wave (lock step makes
        // a wave as slow as its slowest instance) and the batch is several times what is resident.  Option "tile_dyn": 0 never, 1 always.
        SolveKernel ks = (ub && te->kub) ? te->kub : te->k, kd = (ub && te->kubdyn) ? te->kubdyn : te->kdyn;
        int resident = 0;
        bool dyn = kd != nullptr && b->tile_dyn_opt != 0;
        if (dyn) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kd), 64, 0) != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); per_cu = 4; }
            resident = per_cu * b->num_cus;
            if (b->tile_dyn_opt < 0 && (ipw < 2 || (long)grid < 4L * resident || b->grid_waves_per_cu > 0)) dyn = false;
        }
        if (dyn) {
            if (!b->d_work_counter) HIP_TRY(b, hipMalloc(reinterpret_cast<void**>(&b->d_work_counter), sizeof(int)));
            HIP_TRY(b, hipMemsetAsync(b->d_work_counter, 0, sizeof(int), b->stream));
            a.work_counter = b->d_work_counter;
            hipLaunchKernelGGL(kd, dim3(std::min(grid, resident)), dim3(64), 0, b->stream, a);
        } else {
            hipLaunchKernelGGL(ks, dim3(grid), dim3(64), 0, b->stream, a);
        }
        HIP_TRY(b, hipGetLastError());
        b->last_tile_dyn = dyn;
    }
    b->last_tile_form = te ? (te->W * 1000000 + te->R * 1000 + te->lm) : -1;      // which tile_dims.txt entry ran (-1: run-time instantiated)
    if (timed) { HIP_TRY(b, hipEventRecord(b->ev_stop[b->timing_n], b->stream)); b->timing_n++; b->timing_left--; }
    if (!dry) {
        b->status_valid = true;
        if (b->d_traj) b->traj_step += steps;        // the window moves one knot per MPC step
    }
    return TINY_OK;
}

static bool linear_active(const TinyBatch* b) {
    return b->set.en_state_linear || b->set.en_input_linear || b->set.en_tv_state_linear || b->set.en_tv_input_linear;
}
// LIN template value of the register-resident linear-constraint variant that can serve the current settings, 0 if none
static int lin_variant(const TinyBatch* b) {
    if (!linear_active(b) || !has_regs(b) || b->force_general) return 0;
    if (b->variant_jit_failed && b->debug) return 0;                   // LIN x debug needs hipRTC; without it: coverage kernel
    if (lin_kmax(b) == 0) return 0;                                    // too many half-spaces per knot: coverage kernel
    if (lin_kmax(b) > LIN_KMAX && (b->no_jit || b->variant_jit_failed)) return 0;
    const int lv = ((b->set.en_state_linear || b->set.en_input_linear) ? 1 : 0) | ((b->set.en_tv_state_linear || b->set.en_tv_input_linear) ? 2 : 0);
    // a LEAN shape (kernel_entry.hpp KERNELS_LEAN) carries no half-space variant compiled in: without hipRTC the coverage kernel serves it
    if (b->variant_jit_failed && b->kernel && !b->hetero && !b->kernel->klin[soc_active(b) ? 1 : 0][lv]) return 0;
    return lv;
}
// The compiled-in instantiation of the one-row kernel that serves a box / cone launch with these debug outputs and FMA block
// form, nullptr if the shape's set does not hold it (LEAN shapes: only <box, no debug, mode 2> and the UB form)
static SolveKernel compiled_in_plain_variant(const TinyBatch* b, bool soc, bool dbg, int mode) {
    if (!b->kernel) return nullptr;
    if (!dbg && mode == 2 && b->bounds_uniform && b->use_ub) {
        if (SolveKernel k = soc ? b->kernel->kubsoc : b->kernel->kub) return k;
    }
    return b->kernel->k[soc ? 1 : 0][dbg ? 1 : 0][mode];
}
// cones of an ENABLED family share rows: sequential projections (admm.cpp:111-135), coverage kernel only
static bool cones_overlap(const TinyBatch* b) {
    return (b->set.en_state_soc && b->cones_overlap_x) || (b->set.en_input_soc && b->cones_overlap_u);
}
static bool use_general(const TinyBatch* b) {
    if (cones_overlap(b)) return true;
    if (b->adaptive) return false;                    // adaptive rho lives on the one-row kernel only (launch_solve refuses the rest)
    if (linear_active(b)) return lin_variant(b) == 0;
    if (b->hetero) return false;
    // a variant outside the compiled-in set of a LEAN shape that hipRTC could not make either (no_jit, no hipRTC on the box, a
    // compile error): the coverage kernel serves cone / debug / dpp-mode launches exactly as it does for shapes outside kernel_dims.txt
    if (b->variant_jit_failed && b->kernel && !compiled_in_plain_variant(b, soc_active(b), b->debug, (b->dpp_mode >= 0 && b->dpp_mode <= 2) ? b->dpp_mode : 0)) return true;
    return !has_regs(b) || b->force_general;
}

// Tables of the coverage kernel (general_kernel.hip.h): row-major [row][nz+1] matrices + vectors + constraints.
static void build_general_tables(TinyBatch* b) {
    const int nx = b->nx, nu = b->nu, N = b->N, nz = nx + nu, ld = nz + 1;
    const Cache& c = b->cache;
    GeneralArgs& g = b->gargs;
    int off = 0;
    auto take = [&](int n) { int o = off; off += n; return o; };
    g.o_mb = take(nz * ld); g.o_mf1 = take(nz * ld); g.o_mf2 = take(nz * ld); g.o_pt = take(nz * ld);
    g.o_cb = take(nz); g.o_cf = take(nz); g.o_qr = take(nz);
    g.o_lo = take(N * nz); g.o_hi = take(N * nz);
    g.o_sc = take(2 * (int)b->Acx.size() + 2); g.o_ic = take(2 * (int)b->Acu.size() + 2);
    g.o_ax = take(b->nsl * nx + 1); g.o_bx = take(b->nsl + 1); g.o_au = take(b->nil * nu + 1); g.o_bu = take(b->nil + 1);
    g.o_tax = take(N * b->ntsl * nx + 1); g.o_tbx = take(N * b->ntsl + 1);
    g.o_tau = take((N - 1) * b->ntil * nu + 1); g.o_tbu = take((N - 1) * b->ntil + 1);
    std::vector<double>& t = b->h_gtab;
    t.assign(off, 0.0);
    Mat QBt = c.Quu_inv * transpose(b->B);
    Mat QBPf = c.Quu_inv * c.BPf;
    for (int j = 0; j < nx; ++j) {
        for (int k = 0; k < nx; ++k) {
            t[g.o_mb + j * ld + k] = c.AmBKt(j, k);
            t[g.o_mf1 + j * ld + k] = b->A(j, k);
            t[g.o_pt + j * ld + k] = c.Pinf(k, j);
        }
        for (int m = 0; m < nu; ++m) {
            t[g.o_mb + j * ld + nx + m] = -c.Kinf(m, j);
            t[g.o_mf2 + j * ld + nx + m] = b->B(j, m);
        }
        t[g.o_cb + j] = c.APf(j, 0); t[g.o_cf + j] = b->f(j, 0); t[g.o_qr + j] = b->Qw[j];
    }
    for (int a = 0; a < nu; ++a) {
        const int j = nx + a;
        for (int k = 0; k < nx; ++k) { t[g.o_mb + j * ld + k] = QBt(a, k); t[g.o_mf1 + j * ld + k] = -c.Kinf(a, k); }
        for (int m = 0; m < nu; ++m) t[g.o_mb + j * ld + nx + m] = c.Quu_inv(a, m);
        t[g.o_cb + j] = QBPf(a, 0); t[g.o_qr + j] = b->Rw[a];
    }
    const double inf = std::numeric_limits<double>::infinity();
    for (int e = 0; e < N * nz; ++e) { t[g.o_lo + e] = -inf; t[g.o_hi + e] = inf; }
    if (b->set.en_state_bound && b->have_bounds)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < nx; ++j) { t[g.o_lo + i * nz + j] = b->x_min[(size_t)i * nx + j]; t[g.o_hi + i * nz + j] = b->x_max[(size_t)i * nx + j]; }
    if (b->set.en_input_bound && b->have_bounds)
        for (int i = 0; i < N - 1; ++i)
            for (int a = 0; a < nu; ++a) { t[g.o_lo + i * nz + nx + a] = b->u_min[(size_t)i * nu + a]; t[g.o_hi + i * nz + nx + a] = b->u_max[(size_t)i * nu + a]; }
    for (size_t k = 0; k < b->Acx.size(); ++k) { t[g.o_sc + 2 * k] = b->Acx[k]; t[g.o_sc + 2 * k + 1] = b->cx[k]; }
    for (size_t k = 0; k < b->Acu.size(); ++k) { t[g.o_ic + 2 * k] = b->Acu[k]; t[g.o_ic + 2 * k + 1] = b->cu[k]; }
    std::copy(b->Alin_x.begin(), b->Alin_x.end(), t.begin() + g.o_ax); std::copy(b->blin_x.begin(), b->blin_x.end(), t.begin() + g.o_bx);
    std::copy(b->Alin_u.begin(), b->Alin_u.end(), t.begin() + g.o_au); std::copy(b->blin_u.begin(), b->blin_u.end(), t.begin() + g.o_bu);
    std::copy(b->tvA_x.begin(), b->tvA_x.end(), t.begin() + g.o_tax); std::copy(b->tvb_x.begin(), b->tvb_x.end(), t.begin() + g.o_tbx);
    std::copy(b->tvA_u.begin(), b->tvA_u.end(), t.begin() + g.o_tau); std::copy(b->tvb_u.begin(), b->tvb_u.end(), t.begin() + g.o_tbu);
}

static int ensure_kpi(TinyBatch* b, double** p) {
    if (*p) return TINY_OK;
    const size_t kpi_bytes = (size_t)b->batch * b->N * (b->nx + b->nu) * sizeof(double);
    HIP_TRY(b, hipMalloc(p, kpi_bytes));
    HIP_TRY(b, hipMemsetAsync(*p, 0, kpi_bytes, b->stream));
    return TINY_OK;
}

// phase = 0: the whole solve; > 0: one phase of the iteration (PHASE_* of general_kernel.hip.h) on the records as they are
static int launch_general(TinyBatch* b, int phase = 0) {
    if (b->nx + b->nu > 32) return fail(b, TINY_ERR_UNSUPPORTED, "nx + nu = %d > 32", b->nx + b->nu);
    if (phase == 0) {
        if (b->steps_per_launch > 1) return fail(b, TINY_ERR_UNSUPPORTED, "steps_per_launch needs a register-resident kernel instantiation");
        if (b->d_traj || b->reset_duals || b->one_shot) return fail(b, TINY_ERR_UNSUPPORTED, "reference-trajectory windows / reset_duals / one_shot need a register-resident kernel instantiation");
    }
    double** need[] = {&b->d_dbg_qr, &b->d_dbg_pd, &b->d_lslack, &b->d_ldual, &b->d_tlslack, &b->d_tldual};
    const bool want[] = {true, true, b->set.en_state_linear || b->set.en_input_linear, b->set.en_state_linear || b->set.en_input_linear,
                         b->set.en_tv_state_linear || b->set.en_tv_input_linear, b->set.en_tv_state_linear || b->set.en_tv_input_linear};
    for (int i = 0; i < 6; ++i)
        if (want[i]) { if (int rc = ensure_kpi(b, need[i])) return rc; }
    if (b->tab_dirty || b->h_gtab.empty() || phase > 0) {
        build_general_tables(b);
        if (b->gtab_doubles < b->h_gtab.size()) {
            if (b->d_gtab) hipFree(b->d_gtab);
            HIP_TRY(b, hipMalloc(&b->d_gtab, b->h_gtab.size() * sizeof(double)));
            b->gtab_doubles = b->h_gtab.size();
        }
        HIP_TRY(b, hipMemcpyAsync(b->d_gtab, b->h_gtab.data(), b->h_gtab.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
    }
    GeneralArgs a = b->gargs;
    a.gtab = b->d_gtab; a.x0 = b->d_x0; a.ref = b->d_ref; a.prim = b->d_prim; a.slack = b->d_slack; a.dual = b->d_dual;
    a.slack_prev = b->d_slack_prev; a.cslack = b->d_cslack; a.cdual = b->d_cdual; a.lslack = b->d_lslack; a.ldual = b->d_ldual;
    a.tlslack = b->d_tlslack; a.tldual = b->d_tldual; a.qr = b->d_dbg_qr; a.pd = b->d_dbg_pd;
    a.status = b->d_status; a.resid = b->d_resid; a.accum = b->d_accum; a.x0_next = b->advance_x0 ? b->d_x0 : nullptr;
    a.rho = b->cache.rho; a.tol_pri = b->set.abs_pri_tol; a.tol_dua = b->set.abs_dua_tol;
    a.batch = b->batch; a.max_iter = b->set.max_iter; a.check_termination = b->set.check_termination;
    a.nx = b->nx; a.nu = b->nu; a.N = b->N;
    a.soc_s = b->set.en_state_soc && !b->Acx.empty(); a.soc_i = b->set.en_input_soc && !b->Acu.empty();
    a.n_sc = b->set.en_state_soc ? (int)b->Acx.size() : 0; a.n_ic = b->set.en_input_soc ? (int)b->Acu.size() : 0;
    a.lin_s = b->set.en_state_linear; a.lin_i = b->set.en_input_linear;
    a.tlin_s = b->set.en_tv_state_linear; a.tlin_i = b->set.en_tv_input_linear;
    a.nsl = b->nsl; a.nil = b->nil; a.ntsl = b->ntsl; a.ntil = b->ntil;
    const int nz = b->nx + b->nu;
    const size_t lds = (size_t)(4 * nz * (nz + 1) + nz + b->nu + b->nx + 3 * b->N * nz) * sizeof(double);
    if (lds > 160 * 1024) return fail(b, TINY_ERR_UNSUPPORTED, "(nx,nu,N)=(%d,%d,%d) needs %zu B of LDS per instance (> 160 KiB)", b->nx, b->nu, b->N, lds);
    if (phase == 0 && lds > 64 * 1024 && lds > b->general_lds_limit) {
        HIP_TRY(b, hipFuncSetAttribute(reinterpret_cast<const void*>(admm_general_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        b->general_lds_limit = lds;
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 16) per_cu = 16;
    if (per_cu < 1) per_cu = 1;
    int grid = b->batch;
    const long cap = (long)b->num_cus * per_cu;
    if (cap < grid) grid = (int)cap;
    if (phase > 0) {
        if (lds > 64 * 1024 && lds > b->phase_lds_limit) {
            HIP_TRY(b, hipFuncSetAttribute(reinterpret_cast<const void*>(admm_phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            b->phase_lds_limit = lds;
        }
        hipLaunchKernelGGL(admm_phase_kernel, dim3(grid), dim3(64), lds, b->stream, a, phase);
        HIP_TRY(b, hipGetLastError());
        return TINY_OK;
    }
    const bool timed = b->timing_left > 0 && b->timing_n < (int)b->ev_start.size();
    if (timed) HIP_TRY(b, hipEventRecord(b->ev_start[b->timing_n], b->stream));

    hipLaunchKernelGGL(admm_general_kernel, dim3(grid), dim3(64), lds, b->stream, a);
    HIP_TRY(b, hipGetLastError());
    if (timed) { HIP_TRY(b, hipEventRecord(b->ev_stop[b->timing_n], b->stream)); b->timing_n++; b->timing_left--; }
    return TINY_OK;
}

static int upload_tables(TinyBatch* b) {
    if (!b->tab_dirty) return TINY_OK;
    build_tables(b);
    if (b->h_tab.size() > b->d_tab_doubles) {
        HIP_TRY(b, hipStreamSynchronize(b->stream));
        (void)hipFree(b->d_tab);
        b->d_tab = nullptr; b->d_tab_doubles = 0;
        HIP_TRY(b, hipMalloc(&b->d_tab, b->h_tab.size() * sizeof(double)));
        b->d_tab_doubles = b->h_tab.size();
    }
    HIP_TRY(b, hipMemcpyAsync(b->d_tab, b->h_tab.data(), b->h_tab.size() * sizeof(double), hipMemcpyHostToDevice,
                              b->stream));
    // h_tab is pageable: the copy above is staged synchronously, so reusing h_tab later is safe
    b->tab_dirty = false;
    return TINY_OK;
}

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
static double wave_iteration_us(int nx, int nu, int N, int wps) {
    const double S = (double)nx * N + (double)nu * (N - 1);
    const double fl = 4.0 * S + 2.0 * nx * nx + 3.0 * nx + (N - 1.0) * (4.0 * nx * nx + 8.0 * nx * nu + 2.0 * nu * nu + 4.0 * nu + 5.0 * nx) + 11.0 * S;
    return 4.0 * fl * wps / (76.8e3 * (wps == 2 ? 0.75 : 0.45));
}
// the K (multiple of check_termination) with the smallest predicted time, 0 when a plain launch is within 5 % of it
static int choose_split_for(int nx, int nu, int N, bool soc, int M, int ct, int gr, int num_cus, const unsigned* hist, double* ratio, int* growth_out = nullptr) {
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
static int choose_split(const TinyBatch* b, const unsigned* hist, double* ratio) {
    return choose_split_for(b->nx, b->nu, b->N, soc_active(b), b->set.max_iter, std::max(1, b->set.check_termination), b->repack_growth >= 2 ? b->repack_growth : 0,
                            b->num_cus, hist, ratio, &const_cast<TinyBatch*>(b)->auto_growth);
}

// ---- adaptive rho: per-instance cache state + the lane tables of the adaptation step -----------------------------------
static __global__ void broadcast_vec_kernel(double* __restrict__ dst, const double* __restrict__ src, long n, int per) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i % per];
}

// every instance's cache state <- the family's cache (rho, Kinf, Pinf, C1 = Quu_inv, C2 = AmBKt: tiny_api.cpp:375-376)
static int adaptive_fresh_state(TinyBatch* b) {
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

static int ensure_adaptive(TinyBatch* b, bool need_tables = true) {
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

int launch_projection(int which, double* v, const double* a, int n, float mu, double bb) {
    if (which == 0) hipLaunchKernelGGL(project_soc_kernel, dim3(1), dim3(64), 0, nullptr, v, n, mu);
    else hipLaunchKernelGGL(project_hyperplane_kernel, dim3(1), dim3(64), 0, nullptr, v, a, n, bb);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return TINY_ERR_HIP;
    return TINY_OK;
}

bool soc_active(const TinyBatch* b) {
    return (b->set.en_state_soc && !b->Acx.empty()) || (b->set.en_input_soc && !b->Acu.empty());
}

// histogram of the iteration counts the solve just enqueued leaves in d_status -> pinned host memory, asynchronously (hist_ev)
static int enqueue_iteration_histogram(TinyBatch* b) {
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
static int ensure_regroup_buffers(TinyBatch* b, bool second_stream) {
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
static int enqueue_regroup_sort(TinyBatch* b, hipStream_t st, int half, int first, int count) {
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
static int enqueue_repack_sort(TinyBatch* b, const int* list, const int* count) {
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
static int enqueue_lockstep_estimate(TinyBatch* b) {
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
static void read_lockstep_estimate(TinyBatch* b) {
    b->ls_pending = false;
    b->lockstep_ratio = b->h_ls[1] > 0ull ? (double)b->h_ls[0] / (double)b->h_ls[1] : 1.0;
    if (b->regroup_verdict == 0) b->regroup_verdict = b->lockstep_ratio >= 1.05 ? 1 : -1;
}
// stretches of `steps` MPC steps: `lead` steps first (0: none), then K steps each; a short remainder joins the stretch before it
static std::vector<int> regroup_stretches(int steps, int K, int lead) {
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
constexpr int REGROUP_AUTO_MIN_STEPS = 16, REGROUP_AUTO_MIN_BATCH = 4096;
static int regroup_auto_k(int steps) { return std::max(8, (steps + 3) / 4); }
// the two-stream form (halves of the batch half a stretch out of step) only makes sense when more than one stretch is left after the
// lead step: the condition the launch and tiny_step_regroup_plan share
static bool regroup_two_streams_apply(int steps, int lead, int K) { return steps - lead > K; }

// index lists + per-stage counters of the split solve: [stage] list lengths, [32 + stage] tile counters
static int ensure_repack_buffers(TinyBatch* b) {
    if (b->d_repack_index && b->d_repack_count) return TINY_OK;
    if (!b->d_repack_index) HIP_TRY(b, hipMalloc(&b->d_repack_index, 2 * (size_t)b->batch * sizeof(int)));
    if (!b->d_repack_count) HIP_TRY(b, hipMalloc(&b->d_repack_count, 2 * 32 * sizeof(int)));
    return TINY_OK;
}

// ---- launch_solve in three parts -----------------------------------------------------------------------------
//   DECIDE   which launch form this solve takes: from the options and from the PLAN, i.e. what earlier solves of the batch left behind
//            (plain or split and its K / stage schedule, the tile kernel's dynamic form for a one-row shape, stretches of MPC steps);
//            the plan can be exported and imported (tiny_batch_get_plan / tiny_batch_set_plan) so that a process need not re-probe
//   ENQUEUE  the launches of that form (enqueue_split_solve / enqueue_plain_or_stretches / launch_tile / launch_general)
//   LEARN    while a question is open the solve is timed and leaves its iteration histogram behind (learn_from_probe reads them
//            when the NEXT solve of the batch is decided; a solve never waits)
enum { MAX_STAGES = 32 };
struct OneRowLaunch {                                // one launch_solve call on the one-row kernel
    SolveArgs a;
    SolveKernel k = nullptr;                         // the compiled-in instantiation, or
    hipFunction_t jit_fn = nullptr;                  // ... the one hipRTC made
    JitKey jk;
    int steps = 1, ipw = 4, grid = 0;
    bool soc = false;
    // the decision
    bool split_ok = false, auto_split = false, growth_probe = false, auto_probe = false;
    int cap = 0;                                     // > 0: split solve with first stage K = cap
};

// arguments, variant and kernel of the launch (the tables are uploaded here).  `done`: the launch was handed to another path (a
// variant that could not be instantiated: coverage kernel) and the value returned is that launch's
static int build_one_row_launch(TinyBatch* b, const bool zero_state, OneRowLaunch& L, bool& done) {
    b->h_gtab.clear();
    if (int rc = upload_tables(b)) return rc;
    const bool soc = L.soc = soc_active(b);
    SolveArgs& a = L.a;
    a.arho = a.aK = a.aP = a.aC1 = a.aC2 = nullptr; a.atab = nullptr; a.arho_min = a.arho_max = 0.0; a.aclip = 0; a.ref_shared = 0; a.work_counter = nullptr; a.reverse = 0;
    a.index = nullptr; a.count = nullptr; a.iter_base = 0; a.next_index = nullptr; a.next_count = nullptr; a.perm = nullptr; a.perm_count = 0;
    a.tab = b->d_tab; a.x0 = b->d_x0; a.ref = b->d_ref; a.prim = b->d_prim; a.slack = b->d_slack;
    a.dual = b->d_dual; a.slack_prev = b->d_slack_prev; a.cslack = b->d_cslack; a.cdual = b->d_cdual;
    a.status = b->d_status; a.resid = b->d_resid;
    const int steps = L.steps = b->steps_per_launch > 1 ? b->steps_per_launch : 1;
    a.steps = steps;
    a.x0_next = (b->advance_x0 || steps > 1) ? b->d_x0 : nullptr;     // fused steps imply the plant step
    a.iter_log = nullptr; a.u0_log = nullptr;
    a.het_tabs = b->hetero ? b->d_het_tabs : nullptr;
    a.traj = b->d_traj; a.traj_offsets = b->d_traj_offsets; a.traj_points = b->traj_points;
    a.traj_step0 = (int)b->traj_step; a.reset_duals = b->reset_duals ? 1 : 0;
    a.cold = (b->one_shot || (zero_state && b->auto_cold)) ? 1 : 0;
    a.ref_shared = (b->share_ref && b->xref_shared && b->uref_shared) ? 1 : 0;
    a.store_mask = b->one_shot == 2 ? 1 : (b->one_shot == 1 ? 3 : 31);
    // "store_primal" = 0: work->x|u is not written back.  A cone / half-space slack is initialised from it by the next
    // solve (admm.cpp:352-374) and the debug outputs belong to it, so those launches keep the store.
    if (b->store_primal != 1 && !b->one_shot && !soc && !lin_variant(b) && !b->debug) a.store_mask = (a.store_mask & ~1) | (b->store_primal == 2 ? 32 : 0);
    if (steps > 1 && b->step_log) {
        if (int rc = ensure_step_logs(b, steps)) return rc;
        a.iter_log = b->d_iter_log; a.u0_log = b->d_u0_log;
    }
    a.dbg_qr = b->debug ? b->d_dbg_qr : nullptr;
    a.dbg_pd = b->debug ? b->d_dbg_pd : nullptr;
    a.accum = b->d_accum;
    a.rho = b->cache.rho; a.tol_pri = b->set.abs_pri_tol; a.tol_dua = b->set.abs_dua_tol;
    a.batch = b->batch; a.max_iter = b->set.max_iter; a.check_termination = b->set.check_termination;
    const int tiles = (b->batch + 3) / 4;
    int& grid = L.grid;
    grid = tiles;
    if (b->grid_waves_per_cu > 0) {
        const long cap = (long)b->num_cus * b->grid_waves_per_cu;
        if (cap < grid) grid = (int)cap;
    }
    // the variant this launch needs: <SOC, DBG, MODE, LIN, HET>.  kernel_dims.txt shapes carry the common ones compiled in
    // (every soc x debug x mode; LIN and HET without debug at mode 2); anything else -- an unseen shape, or debug outputs
    // together with half-spaces / per-instance data, or both of those -- is instantiated now (jit.hpp).
    const int mode = (b->dpp_mode >= 0 && b->dpp_mode <= 2) ? b->dpp_mode : 0;
    JitKey& jk = L.jk;
    jk = JitKey{b->nx, b->nu, b->N, soc ? 1 : 0, b->debug ? 1 : 0, mode, 0, b->hetero ? 1 : 0, LIN_KMAX, 0};
    a.lslack = a.ldual = a.tlslack = a.tldual = nullptr;
    a.n_lin = a.n_tlin = 0;
    if (const int lv = lin_variant(b)) {
        if (lv & 1) { if (int rc = ensure_kpi(b, &b->d_lslack)) return rc; if (int rc = ensure_kpi(b, &b->d_ldual)) return rc; }
        if (lv & 2) { if (int rc = ensure_kpi(b, &b->d_tlslack)) return rc; if (int rc = ensure_kpi(b, &b->d_tldual)) return rc; }
        a.lslack = b->d_lslack; a.ldual = b->d_ldual; a.tlslack = b->d_tlslack; a.tldual = b->d_tldual;
        // half-spaces applied per knot: the larger count of the ENABLED families (a disabled family's rows are inert)
        a.n_lin = std::max(b->set.en_state_linear ? b->nsl : 0, b->set.en_input_linear ? b->nil : 0);
        a.n_tlin = std::max(b->set.en_tv_state_linear ? b->ntsl : 0, b->set.en_tv_input_linear ? b->ntil : 0);
        jk.lin = lv;
        jk.kmax = lin_kmax(b);
    }
    if (b->adaptive) {
        if (b->hetero || jk.lin || b->one_shot || b->repack_after > 0)
            return fail(b, TINY_ERR_UNSUPPORTED, "adaptive rho does not combine with heterogeneous data / half-spaces / one_shot / repack_after");
        if (int rc = ensure_adaptive(b)) return rc;
        a.arho = b->d_arho; a.aK = b->d_aK; a.aP = b->d_aP; a.aC1 = b->d_aC1; a.aC2 = b->d_aC2; a.atab = b->d_atab;
        a.arho_min = b->adaptive_min; a.arho_max = b->adaptive_max; a.aclip = b->adaptive_clip ? 1 : 0;
        b->astate_fresh = false;
        jk.adapt = 1;
    }
    if (jk.lin || jk.het || jk.adapt) jk.mode = 2;   // those variants exist on the single-chain FMA blocks only
    SolveKernel& k = L.k;
    k = nullptr;
    if (b->kernel) {
        if (jk.adapt) k = jk.soc ? nullptr : b->kernel->kadapt[jk.dbg];
        else if (!jk.lin && !jk.het && !jk.dbg && jk.mode == 2 && b->bounds_uniform && b->use_ub) k = jk.soc ? b->kernel->kubsoc : b->kernel->kub;
        else if (!jk.lin && !jk.het) k = b->kernel->k[jk.soc][jk.dbg][jk.mode];
        else if (jk.lin && !jk.het && !jk.dbg && jk.kmax == LIN_KMAX) k = b->kernel->klin[jk.soc][jk.lin];
        else if (jk.het && !jk.lin && !jk.dbg) k = b->kernel->khet[jk.soc];
    }
    // HALF rows (nx+nu <= 8): the plain box launch of a compiled-in shape takes the form that puts two instances into a DPP row --
    // eight per wave (option "half_rows": -1 / 1 on where the form exists, 0 off).  Bit-identical to the one-row form.
    int& ipw = L.ipw;
    ipw = 4;
    if (k && b->kernel && b->half_rows != 0 && (k == b->kernel->kub || k == b->kernel->k[0][0][2])) {
        SolveKernel kh = b->kernel->khalf[k == b->kernel->kub ? 1 : 0];
        if (kh) { k = kh; ipw = 8; }
    }
    b->last_half = ipw == 8;
    if (ipw == 8) {
        grid = (b->batch + 7) / 8;
        if (b->grid_waves_per_cu > 0) grid = (int)std::min<long>(grid, (long)b->num_cus * b->grid_waves_per_cu);
    }
    hipFunction_t& jit_fn = L.jit_fn;
    jit_fn = nullptr;
    if (!k) {
        std::string why;
        if (b->no_jit) why = "run-time instantiation is switched off (no_jit)";
        else jit_fn = jit_solve_kernel(jk, &why);
        if (!jit_fn) {
            if (b->kernel) {
                // the shape's compiled-in set does not hold this variant (half-spaces + debug outputs; any cone / debug / dpp-mode /
                // half-space variant of a LEAN shape) and hipRTC could not make it: the coverage kernel can do everything but
                // per-instance data, adaptive rho and the fused / windowed / one-shot launch forms
                const bool general_can = !jk.het && !jk.adapt && steps == 1 && !b->d_traj && !b->reset_duals && !b->one_shot;
                if (general_can && !b->variant_jit_failed) {
                    b->variant_jit_failed = true;
                    b->tab_dirty = true; b->redispatch = true;
                    { done = true; return launch_solve(b); }
                }
                return fail(b, TINY_ERR_UNSUPPORTED, "this combination of cone / debug outputs / half-spaces / per-instance data / launch form needs hipRTC: %s", why.c_str());
            }
            if (b->adaptive) return fail(b, TINY_ERR_UNSUPPORTED, "adaptive rho kernel for (nx,nu,N)=(%d,%d,%d) could not be instantiated: %s", b->nx, b->nu, b->N, why.c_str());
            b->jit_failed = true;                    // has_regs() turns false: the coverage kernel takes over
            b->tab_dirty = true; b->redispatch = true;
            if (b->hetero || b->steps_per_launch > 1 || b->d_traj || b->reset_duals || b->one_shot)
                return fail(b, TINY_ERR_UNSUPPORTED, "no register-resident kernel for (nx,nu,N)=(%d,%d,%d): %s", b->nx, b->nu, b->N, why.c_str());
            { done = true; return launch_solve(b); }
        }
    }
    return TINY_OK;
}

// LEARN: the clock's word on the previous eligible solve of this batch (its events and its iteration histogram have arrived)
static void learn_from_probe(TinyBatch* b, const int max_iter, const bool auto_split) {
    struct { int max_iter; } a = {max_iter};
    if (auto_split && b->hist_pending && hipEventQuery(b->hist_ev) == hipSuccess) {
        b->hist_pending = false;
        // the clock's word on the previous eligible solve: microseconds per instance-iteration, plain or split
        float ms = 0.0f;
        double iters = 0.0;
        for (int i = 0; i < TinyBatch::HIST_BINS; ++i) iters += (double)i * b->h_hist[i];
        if (++b->auto_probes > 1 && iters > 0.0 && hipEventElapsedTime(&ms, b->auto_ev0, b->auto_ev1) == hipSuccess && ms > 0.0f) {
            const double rate = (double)ms / iters;
            if (b->probe_was_tile) {
                // the dynamic slot form of the tile kernel (one-row layout) against the best the one-row kernel did on this batch
                b->tile_rate = rate;
                const double best = (b->auto_verdict == 1 && b->auto_split_rate > 0.0) ? b->auto_split_rate : b->auto_plain_rate;
                if (best > 0.0) b->tile_verdict = rate < 0.97 * best ? 1 : -1;
            } else if (b->probe_was_growth) {
                // the other stage schedule of a kept split (K, 4K, ... against K, 2K, ...): the cost model ranks them, the clock decides
                if (b->auto_split_rate > 0.0 && rate < 0.97 * b->auto_split_rate) { b->auto_growth = b->growth_alt; b->auto_split_rate = rate; }
                b->growth_verdict = 1;
            } else {
                if (b->auto_last_cap > 0) b->auto_split_rate = rate; else b->auto_plain_rate = rate;
                if (b->auto_last_cap > 0 && b->auto_plain_rate > 0.0 && b->auto_verdict == 0)
                    b->auto_verdict = b->auto_split_rate < 0.97 * b->auto_plain_rate ? 1 : -1;
            }
        }
        if (b->probe_was_growth) b->growth_verdict = 1;           // (asked once, whatever became of the reading)
        b->probe_was_tile = false; b->probe_was_growth = false;
        if (b->auto_verdict == 0) {                   // (a kept split keeps its K; a rejected one stays rejected until the options change)
            b->auto_cap = choose_split(b, b->h_hist, &b->auto_gain);
            b->auto_cap_max_iter = a.max_iter;
            b->hist_copy.assign(b->h_hist, b->h_hist + TinyBatch::HIST_BINS);
        }
    }
}

// DECIDE + ENQUEUE, the tile kernel's dynamic form for a one-row shape.  `done`: this solve ran there
static int try_tile_alternative(TinyBatch* b, const OneRowLaunch& L, bool& done) {
    const bool auto_split = L.auto_split, soc = L.soc;
    const JitKey& jk = L.jk;
    // The same batch on the tile kernel's dynamic slot form (its one-row layout, tile_dims.txt): persistent waves whose rows take
    // the next instance off a device-wide counter the moment they are free.  It wins where solves are long and their iteration
    // counts spread (3-32 % on 14 of the 16 N = 10 / 30 config-5 cells) and loses where they are short (config 3: 2.4x), so the
    // clock decides here as well: once the one-row kernel's own question (plain or split) is settled, ONE eligible solve runs on
    // the dynamic form, timed; it is kept if it beats the one-row kernel's best time per instance-iteration by 3 %.
    const bool tile_alt_ok = auto_split && b->tile && !b->tile_is_jit && b->tile->W <= 1 && b->tile_dyn_opt < 0 && !b->prefer_tile && !soc && !jk.lin &&
                             !jk.het && !jk.adapt && !jk.dbg && !b->d_traj && !b->reset_duals && b->store_primal == 1 && !b->no_tile && b->repack_after < 0;
    if (tile_alt_ok) {
        const bool one_row_settled = b->auto_plain_rate > 0.0 && (b->auto_verdict == -1 || b->auto_cap == 0 || (b->auto_verdict == 1 && (b->growth_verdict != 0 || b->repack_growth >= 2)));
        const bool probe_tile = b->tile_verdict == 0 && one_row_settled && !b->hist_pending;
        if (b->tile_verdict == 1 || probe_tile) {
            if (b->tile_verdict == 1 && ++b->tile_since >= 32) {             // distributions drift: re-open both questions
                b->tile_since = 0; b->tile_verdict = 0; b->auto_verdict = 0; b->growth_verdict = 0; b->auto_plain_rate = 0.0; b->auto_since = 0;
            } else {
                // the first entry of the shape, dynamic slots -- for THIS launch: the caller's options come back on every way out
                struct Restore {
                    TinyBatch* b; int dyn, lm, r;
                    ~Restore() { b->tile_dyn_opt = dyn; b->tile_lm = lm; b->tile_r = r; }
                } restore{b, b->tile_dyn_opt, b->tile_lm, b->tile_r};
                b->tile_dyn_opt = 1; b->tile_lm = -1; b->tile_r = 0;
                if (probe_tile) {
                    // (what a kernel's FIRST launch pays once must not count against it: an empty launch of the same form goes first)
                    if (int rc0 = launch_tile(b, true)) return rc0;
                    if (!b->auto_ev0) { HIP_TRY(b, hipEventCreate(&b->auto_ev0)); HIP_TRY(b, hipEventCreate(&b->auto_ev1)); }
                    HIP_TRY(b, hipEventRecord(b->auto_ev0, b->stream));
                }
                const int rc = launch_tile(b);
                if (rc != TINY_OK) return rc;
                if (probe_tile) {
                    HIP_TRY(b, hipEventRecord(b->auto_ev1, b->stream));
                    if (int rc2 = enqueue_iteration_histogram(b)) return rc2;
                    b->probe_was_tile = true;
                }
                done = true;
                return TINY_OK;
            }
        }
    }
    return TINY_OK;
}

static int enqueue_kernel(TinyBatch* b, const OneRowLaunch& L, const SolveArgs& a, const int g, hipStream_t st) {
    if (!st) st = b->stream;
    if (L.jit_fn) {
        SolveArgs copy = a;
        void* params[] = {&copy};
        HIP_TRY(b, hipModuleLaunchKernel(L.jit_fn, (unsigned)g, 1, 1, 64, 1, 1, 0, st, params, nullptr));
    } else {
        hipLaunchKernelGGL(L.k, dim3(g), dim3(64), 0, st, a);
        HIP_TRY(b, hipGetLastError());
    }
    return TINY_OK;
}

// ENQUEUE, split solve: stage 1 to iteration K = L.cap, follow-up stages over the lists of open instances
static int enqueue_split_solve(TinyBatch* b, OneRowLaunch& L) {
    SolveArgs& a = L.a;
    const int cap = L.cap, grid = L.grid;
    const bool growth_probe = L.growth_probe, soc = L.soc;
    auto launch = [&](const int g, hipStream_t st = nullptr) -> int { return enqueue_kernel(b, L, a, g, st); };
    {
        if (int rc = ensure_repack_buffers(b)) return rc;
        if (b->repack_sort != 0) { if (int rc = ensure_regroup_buffers(b, false)) return rc; }
        b->last_sorted_stages = 0;
        HIP_TRY(b, hipMemsetAsync(b->d_repack_count, 0, 2 * MAX_STAGES * sizeof(int), b->stream));
        const int full = a.max_iter;
        int stage = 0;
        a.max_iter = cap;
        a.next_index = b->d_repack_index; a.next_count = b->d_repack_count;
        if (int rc = launch(grid)) return rc;
        int gr = b->repack_growth >= 2 ? b->repack_growth : std::max(2, b->auto_growth);      // (option 0: the model's schedule, confirmed or overturned by the clock)
        if (growth_probe) { gr = b->auto_growth == 2 ? 4 : 2; b->growth_alt = gr; b->probe_was_growth = true; }
        for (long base = cap; base < full; base *= gr, ++stage) {
            const bool last = gr * base >= full || stage + 2 >= MAX_STAGES;
            a.iter_base = (int)base; a.max_iter = last ? full : (int)(gr * base); a.reset_duals = 0;
            a.cold = 0;                               // (a resumed stage reads the state the stage before it stored)
            a.index = b->d_repack_index + (size_t)(stage & 1) * b->batch; a.count = b->d_repack_count + stage;
            a.next_index = last ? nullptr : b->d_repack_index + (size_t)((stage + 1) & 1) * b->batch;
            a.next_count = last ? nullptr : b->d_repack_count + stage + 1;
            // fewer waves than tiles: each takes its next tile off the stage's counter when it is free (repack_dynamic = 0: fixed grid stride)
            a.work_counter = b->repack_dynamic ? b->d_repack_count + MAX_STAGES + stage : nullptr;
            // "repack_sort": the stage takes its list ordered by residual / tolerance (see batch_impl.hpp) when the stage is long enough
            // to pay for the three small passes (~20 us) -- predicted as in choose_split_for from the histogram the schedule came
            // from, >= 60 us (measured at 131 072 instances: (4,4,10), stages of 100-150 us, gains 4 % from sorting every stage,
            // (12,2,30) 14 %; config 3's follow-up stages, 5 % of the batch for ~25 us, stay as they are); without a histogram: never
            bool sort_stage = b->repack_sort > 0;
            if (b->repack_sort < 0 && b->hist_copy.size() == (size_t)TinyBatch::HIST_BINS) {
                double open = 0.0, depth = 0.0;              // instances that enter the stage; iterations they run inside it, summed
                const long hi = last ? full : gr * base;
                for (long i = base + 1; i < TinyBatch::HIST_BINS; ++i) {
                    open += b->hist_copy[i];
                    depth += (double)b->hist_copy[i] * (double)(std::min<long>(i, hi) - base);
                }
                const int wps = solve_kernel_waves_per_simd(b->nx + b->nu, b->N, soc);
                sort_stage = depth / 4.0 / (b->num_cus * 4.0 * wps) * wave_iteration_us(b->nx, b->nu, b->N, wps) >= 60.0;
                (void)open;
            }
            if (sort_stage) {
                if (int rc = enqueue_repack_sort(b, a.index, a.count)) return rc;
                a.index = b->d_perm;
                b->last_sorted_stages++;
            }
            if (int rc = launch(std::min(grid, b->num_cus * b->repack_waves_per_cu))) return rc;
            if (last) break;
        }
    }
    return TINY_OK;
}

// ENQUEUE, one launch -- or, for a fused closed-loop launch whose rows disagree, stretches of K MPC steps (step_regroup)
static int enqueue_plain_or_stretches(TinyBatch* b, OneRowLaunch& L) {
    SolveArgs& a = L.a;
    const int steps = L.steps, ipw = L.ipw, grid = L.grid;
    const bool soc = L.soc;
    const JitKey& jk = L.jk;
    auto launch = [&](const int g, hipStream_t st = nullptr) -> int { return enqueue_kernel(b, L, a, g, st); };
    {
        // option "launch_order" = 1: successive plain launches walk the batch in alternating directions (SolveArgs::reverse)
        a.reverse = (b->launch_order == 2 || (b->launch_order == 1 && b->order_flip)) ? 1 : 0;
        b->order_flip = !b->order_flip;
        // option "step_regroup": a fused closed-loop launch in stretches of K MPC steps, each over the instances ordered by the
        // iteration count of their last solve (see batch_impl.hpp).  Every stretch is the launch a caller with steps_per_launch = K
        // would have made: same results, bit for bit; a stretch that is not the last one keeps x|u to itself where nothing reads it.
        const bool regroup_ok = steps > 1 && !b->one_shot;
        if (b->ls_pending && hipEventQuery(b->ls_ev) == hipSuccess) read_lockstep_estimate(b);
        const bool regroup_auto = b->step_regroup < 0 && regroup_ok && steps >= REGROUP_AUTO_MIN_STEPS && b->batch >= REGROUP_AUTO_MIN_BATCH;
        if (regroup_auto && b->regroup_verdict != 0 && ++b->regroup_since >= 64) { b->regroup_since = 0; b->regroup_verdict = 0; }   // (batches drift: ask again)
        const int rk = !regroup_ok ? 0 : (b->step_regroup > 0 ? b->step_regroup : ((regroup_auto && b->regroup_verdict == 1) ? regroup_auto_k(steps) : 0));
        b->last_regroup_stretches = 1;
        if (rk > 0 && rk < steps) {
            if (int rc = ensure_regroup_buffers(b, true)) return rc;
            const int mask_all = a.store_mask, cold0 = a.cold;
            int* const ilog = a.iter_log; double* const ulog = a.u0_log;
            const bool keep_primal = soc || jk.lin || b->debug;        // (the next stretch reads x|u back: admm.cpp:352-374)
            const int traj0 = (int)b->traj_step;
            a.reverse = 0;
            int launches = 0;
            // one stretch: steps [done, done + n) of the instances `first` .. `first + count - 1`, on stream `st`
            auto stretch = [&](hipStream_t st, const int half, const int first, const int count, const int done, const int n, const bool sorted) -> int {
                if (sorted) { if (int rc = enqueue_regroup_sort(b, st, half, first, count)) return rc; }
                a.perm = sorted ? b->d_perm + first : nullptr; a.perm_count = count;
                a.steps = n;
                a.traj_step0 = traj0 + done;
                a.iter_log = ilog ? ilog + (size_t)done * b->batch : nullptr;
                a.u0_log = ulog ? ulog + (size_t)done * b->batch * b->nu : nullptr;
                a.cold = launches == 0 ? cold0 : 0;
                a.store_mask = (done + n < steps && !keep_primal) ? (mask_all & ~(1 | 32)) : mask_all;
                ++launches;
                int g = (count + ipw - 1) / ipw;
                if (b->grid_waves_per_cu > 0) g = (int)std::min<long>(g, (long)b->num_cus * b->grid_waves_per_cu);
                return launch(g, st);
            };
            int done0 = 0;
            if (!b->status_valid) {                 // nothing is known about the instances yet: ONE step of all of them tells them apart
                if (int rc = stretch(b->stream, 0, 0, b->batch, 0, 1, false)) return rc;
                done0 = 1;
            }
            const int half0 = ((b->batch / 2 + 7) / 8) * 8;
            if (b->regroup_streams == 2 && b->stream2 && half0 < b->batch && (b->step_regroup > 0 || b->batch >= 2 * REGROUP_AUTO_MIN_BATCH) && regroup_two_streams_apply(steps, done0, rk)) {
                // two halves on two streams, the second one half a stretch out of step with the first
                const int first[2] = {0, half0}, count[2] = {half0, b->batch - half0};
                const std::vector<int> sched[2] = {regroup_stretches(steps - done0, rk, 0), regroup_stretches(steps - done0, rk, (rk + 1) / 2)};
                hipStream_t st[2] = {b->stream, b->stream2};
                HIP_TRY(b, hipEventRecord(b->rg_fork, b->stream));
                HIP_TRY(b, hipStreamWaitEvent(b->stream2, b->rg_fork, 0));
                int done[2] = {done0, done0}, rc2 = TINY_OK;
                for (size_t c = 0; c < std::max(sched[0].size(), sched[1].size()) && rc2 == TINY_OK; ++c)
                    for (int h = 0; h < 2 && rc2 == TINY_OK; ++h)
                        if (c < sched[h].size()) {
                            rc2 = stretch(st[h], h, first[h], count[h], done[h], sched[h][c], true);
                            done[h] += sched[h][c];
                        }
                // (whatever happened: the batch's stream waits for what the second one was given, so that nothing the caller enqueues
                // next can overtake it)
                HIP_TRY(b, hipEventRecord(b->rg_join, b->stream2));
                HIP_TRY(b, hipStreamWaitEvent(b->stream, b->rg_join, 0));
                if (rc2 != TINY_OK) return rc2;
            } else {
                int done = done0;
                for (const int n : regroup_stretches(steps - done0, rk, 0)) {
                    if (int rc = stretch(b->stream, 0, 0, b->batch, done, n, true)) return rc;
                    done += n;
                }
            }
            a.perm = nullptr;
            b->last_regroup_stretches = launches;
        } else {
            if (int rc = launch(grid)) return rc;
        }
        if (regroup_auto && b->regroup_verdict == 0 && !b->ls_pending) { if (int rc = enqueue_lockstep_estimate(b)) return rc; }
    }
    return TINY_OK;
}

int launch_solve(TinyBatch* b) {
    // records_zero: tiny_batch_reset (or tiny_batch_setup) zeroed every warm-start record and nothing has written one since.  The
    // one-row kernel then takes its state as zero WITHOUT reading it (SolveArgs::cold) -- bit-identical, 3 of the 4 record reads of a
    // solve saved (the first solve after a reset: BASELINE configs 3 and 5, every cold start).  Any launch ends that knowledge.
    const bool zero_state = b->records_zero;
    b->records_zero = false;
    if (b->tab_dirty) {                              // something the tables are built from has changed since the last launch
        b->tab_gen++;
        if (!b->redispatch) b->tile_soc_failed = false;   // (a failed variant instantiation is retried when the CALLER changed something)
    }
    b->redispatch = false;
    if (cones_overlap(b) && (b->hetero || b->adaptive || b->d_traj || b->one_shot || b->steps_per_launch > 1))
        return fail(b, TINY_ERR_UNSUPPORTED, "overlapping cones run on the coverage kernel: no per-instance data, adaptive rho, reference window, one-shot or fused steps with them");
    if (b->hetero && !use_tile(b) && (!has_regs(b) || (linear_active(b) && lin_variant(b) == 0)))
        return fail(b, TINY_ERR_UNSUPPORTED, "heterogeneous problem data needs a register-resident kernel (the one-row kernel with at most 4 half-spaces per knot and family, or the tile kernel's per-instance form)");
    if (b->adaptive && !has_regs(b))
        return fail(b, TINY_ERR_UNSUPPORTED, "adaptive rho needs the register-resident kernel (nx+nu <= 16, horizon within the register file)");
    const int path = use_tile(b) ? 1 : (use_general(b) ? 2 : 0);
    if (path != b->last_path) { b->tab_dirty = true; b->last_path = path; }   // each path has its own table layout
    if (path == 1) {
        const int rc = launch_tile(b);
        if (rc == TINY_OK) b->tab_dirty = false;
        return rc;
    }
    if (use_general(b)) {
        const int rc = launch_general(b);
        if (rc == TINY_OK) b->tab_dirty = false;
        return rc;
    }
    OneRowLaunch L;
    bool done = false;
    {
        const int rc = build_one_row_launch(b, zero_state, L, done);
        if (rc != TINY_OK || done) return rc;
    }
    SolveArgs& a = L.a;
    const int steps = L.steps;
    // ---- DECIDE
    // "repack_after" = -1 (the default): K comes from the iteration histogram of the previous eligible solve of this batch
    // (collected asynchronously; a solve never waits for it) through the cost model above -- a batch whose iteration
    // counts are uniform gets K = 0, i.e. the plain launch.
    const bool split_ok = L.split_ok = steps == 1 && !a.x0_next && !b->one_shot && !b->adaptive && a.check_termination > 0 && a.max_iter >= 16;
    const bool auto_split = L.auto_split = b->repack_after < 0 && split_ok && b->batch >= 8192;
    if (auto_split) {                                // (not inside a timed probe: the first split solve's clock reading must not pay for a hipMalloc)
        if (int rc = ensure_repack_buffers(b)) return rc;
        if (b->repack_sort != 0) { if (int rc = ensure_regroup_buffers(b, false)) return rc; }
    }
    learn_from_probe(b, a.max_iter, auto_split);
    {
        const int rc = try_tile_alternative(b, L, done);
        if (rc != TINY_OK || done) return rc;
    }
    b->last_tile_dyn = false;                        // (this launch runs on the one-row kernel)
    const bool timed = b->timing_left > 0 && b->timing_n < (int)b->ev_start.size();
    if (timed) HIP_TRY(b, hipEventRecord(b->ev_start[b->timing_n], b->stream));
    if (auto_split && b->auto_verdict != 0 && ++b->auto_since >= 32) {      // distributions drift: ask the clock again now and then
        b->auto_since = 0; b->auto_verdict = 0; b->growth_verdict = 0; b->auto_plain_rate = 0.0; b->tile_verdict = 0;
    }
    // this solve is timed and leaves its iteration histogram behind -- while the question is open; a decided batch launches
    // without the two event records (each costs the next launch a dispatch bubble) and without the histogram pass
    // a kept split asks ONE more question: the other stage schedule, timed like the split itself was
    const bool growth_probe = L.growth_probe = auto_split && !b->hist_pending && b->auto_verdict == 1 && b->growth_verdict == 0 && b->repack_growth < 2 &&
                              b->auto_cap > 0 && b->auto_cap_max_iter == a.max_iter && b->auto_split_rate > 0.0;
    const bool auto_probe = L.auto_probe = (auto_split && !b->hist_pending && b->auto_verdict == 0) || growth_probe;
    if (auto_probe) {
        if (!b->auto_ev0) { HIP_TRY(b, hipEventCreate(&b->auto_ev0)); HIP_TRY(b, hipEventCreate(&b->auto_ev1)); }
        HIP_TRY(b, hipEventRecord(b->auto_ev0, b->stream));
    }
    // split solve (repack_after = K): the launch stops at iteration K and lists the instances it leaves open (the kernel's
    // epilogue appends them, one atomic per wave that has any); a launch over that list carries on to 2K, the next one to 4K,
    // ... max_iter (repack_growth = 2; 4: K, 4K, 16K, ...; 0 = what the cost model picked) -- four open instances per wave at every stage, and within a stage nearly all of them run the same number
    // of iterations.  K is a multiple of check_termination so that the termination countdown of every stage is in phase.
    // Two index lists alternate; every stage has its own counter, all of them zeroed by one memset.
    int& cap = L.cap;
    cap = b->repack_after > 0 ? b->repack_after
            : ((auto_split && b->auto_cap_max_iter == a.max_iter && b->auto_verdict >= 0 && b->auto_plain_rate > 0.0) ? b->auto_cap : 0);
    if (a.check_termination > 1) cap -= cap % a.check_termination;
    // ---- ENQUEUE
    if (int rc = (cap > 0 && cap < a.max_iter && split_ok) ? enqueue_split_solve(b, L) : enqueue_plain_or_stretches(b, L)) return rc;
    // ---- LEARN: what this solve leaves behind for the next decision
    b->status_valid = true;
    if (timed) {
        HIP_TRY(b, hipEventRecord(b->ev_stop[b->timing_n], b->stream));
        b->timing_n++;
        b->timing_left--;
    }
    if (auto_probe) { HIP_TRY(b, hipEventRecord(b->auto_ev1, b->stream)); b->auto_last_cap = (cap > 0 && cap < b->set.max_iter && split_ok) ? cap : 0; }
    if (auto_probe) {                                 // feed the next solve's decision: histogram of THIS solve's iteration counts
        if (int rc = enqueue_iteration_histogram(b)) return rc;
    }
    if (b->d_traj) b->traj_step += steps;            // the window moves one knot per MPC step
    return TINY_OK;
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
        triples[3 * i] = g_kernels[i]->nx; triples[3 * i + 1] = g_kernels[i]->nu; triples[3 * i + 2] = g_kernels[i]->N;
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
    double** kpis[] = {&b->d_ref, &b->d_prim, &b->d_slack, &b->d_dual, &b->d_slack_prev, &b->d_cslack, &b->d_cdual};
    for (double** p : kpis) {
        if (hipMalloc(p, kpi_bytes + kpi_pad) != hipSuccess) return bail(TINY_ERR_HIP);
        if (hipMemsetAsync(*p, 0, kpi_bytes + kpi_pad, b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    }
    b->stage_doubles = (size_t)batch * std::max(nx * N, nu * (N - 1));       // the largest host-layout field (nu > nx happens)
    if (hipMalloc(&b->d_x0, (size_t)batch * nx * sizeof(double)) != hipSuccess) return bail(TINY_ERR_HIP);
    if (hipMemsetAsync(b->d_x0, 0, (size_t)batch * nx * sizeof(double), b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
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
    hipLaunchKernelGGL(riccati_kernel, dim3(grid), dim3(64), lds, b->stream, r);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) return bail(TINY_ERR_HIP);
    std::vector<int> its(B);
    if (hipMemcpy(its.data(), b->d_hiters, B * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return bail(TINY_ERR_HIP);
    for (size_t i = 0; i < B; ++i)
        if (its[i] < 0) { fail(b, TINY_ERR_ARG, "singular R + B'PB for instance %zu", i); return bail(TINY_ERR_ARG); }
    b->hetero = true;
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
    void* bufs[] = {b->d_ref, b->d_prim, b->d_slack, b->d_dual, b->d_slack_prev, b->d_cslack, b->d_cdual, b->d_x0,
                    b->d_stage, b->d_status, b->d_resid, b->d_stats, b->d_tab, b->d_dbg_qr, b->d_dbg_pd, b->d_accum,
                    b->d_iter_log, b->d_u0_log, b->d_lslack, b->d_ldual, b->d_tlslack, b->d_tldual, b->d_gtab, b->d_traj,
                    b->d_traj_offsets, b->d_hA, b->d_hB, b->d_hf, b->d_hQw, b->d_hRw, b->d_hrho, b->d_hK, b->d_hP, b->d_hQuu,
                    b->d_hAmBKt, b->d_hAPf, b->d_hBPf, b->d_het_tabs, b->d_hiters, b->d_ttab, b->d_repack_index, b->d_repack_count,
                    b->d_wire, b->d_arho, b->d_aK, b->d_aP, b->d_aC1, b->d_aC2, b->d_atab, b->d_work_counter, b->d_perm, b->d_rg_bins, b->d_ls};
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
    else if (!strcmp(name, "tile_w")) b->tile_w = (int)value;      // experiments: only entries with this W (0 = half rows, -1 = any)
    else if (!strcmp(name, "tile_lm")) b->tile_lm = (int)value;    // experiments: the tile_dims.txt entry with this LM column
    else if (!strcmp(name, "tile_r")) b->tile_r = (int)value;      // experiments: the tile_dims.txt entry with this R (0: the first that fits)
    else if (!strcmp(name, "step_log")) b->step_log = value != 0;
    else if (!strcmp(name, "reset_duals")) b->reset_duals = value != 0;
    else if (!strcmp(name, "store_primal")) { if (value < 0 || value > 2) return fail(b, TINY_ERR_ARG, "store_primal: 0, 1 or 2"); b->store_primal = (int)value; }
    else if (!strcmp(name, "share_ref")) b->share_ref = value != 0;
    else if (!strcmp(name, "half_rows")) b->half_rows = (int)value;
    else if (!strcmp(name, "launch_order")) { if (value < 0 || value > 2) return fail(b, TINY_ERR_ARG, "launch_order: 0 (ascending), 1 (alternating), 2 (descending)"); b->launch_order = (int)value; }
    else if (!strcmp(name, "step_regroup")) { if (value < -1) return fail(b, TINY_ERR_ARG, "step_regroup: K > 0 (stretches of K MPC steps), 0 (never) or -1 (automatic)"); b->step_regroup = (int)value; b->regroup_verdict = 0; b->regroup_since = 0; b->ls_pending = false; }
    else if (!strcmp(name, "step_regroup_streams")) { if (value < 1 || value > 2) return fail(b, TINY_ERR_ARG, "step_regroup_streams: 1 or 2"); b->regroup_streams = (int)value; }
    else if (!strcmp(name, "auto_cold")) b->auto_cold = value != 0;       // 0: always read the warm-start records, also right after a reset
    else if (!strcmp(name, "uniform_bounds")) b->use_ub = value != 0;
    else if (!strcmp(name, "one_shot")) { if (value < 0 || value > 2) return fail(b, TINY_ERR_ARG, "one_shot: 0, 1 or 2"); b->one_shot = (int)value; }
    else if (!strcmp(name, "repack_after")) { if (value < -1) return fail(b, TINY_ERR_ARG, "repack_after: K > 0, 0 (never) or -1 (automatic)"); b->repack_after = (int)value; b->auto_cap = 0; b->hist_copy.clear(); b->hist_pending = false; b->auto_verdict = 0; b->growth_verdict = 0; b->auto_plain_rate = b->auto_split_rate = 0.0; b->auto_probes = 0; }
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
    if (in->auto_verdict < -1 || in->auto_verdict > 1 || in->tile_verdict < -1 || in->tile_verdict > 1 || in->regroup_verdict < -1 || in->regroup_verdict > 1 ||
        in->auto_cap < 0 || in->auto_cap >= TinyBatch::HIST_BINS || (in->auto_growth != 2 && in->auto_growth != 4) || !(in->auto_plain_rate >= 0.0) || !(in->auto_split_rate >= 0.0))
        return fail(b, TINY_ERR_ARG, "TinyBatchPlan: field out of range");
    HIP_TRY(b, hipSetDevice(b->device));
    // (a plan is advice about launch forms, never about results: every form is bit-identical.  It applies to solves with the
    // max_iter it was made for -- auto_cap_max_iter -- exactly as a plan learnt in this process would)
    b->auto_verdict = in->auto_verdict; b->auto_cap = in->auto_cap; b->auto_cap_max_iter = in->auto_cap_max_iter;
    b->auto_growth = in->auto_growth; b->growth_verdict = in->growth_verdict; b->auto_probes = std::max(2, in->auto_probes);
    b->auto_plain_rate = in->auto_plain_rate; b->auto_split_rate = in->auto_split_rate; b->auto_gain = in->auto_gain;
    b->tile_verdict = in->tile_verdict; b->tile_rate = in->tile_rate;
    b->regroup_verdict = in->regroup_verdict; b->lockstep_ratio = in->lockstep_ratio;
    if (in->hist_valid) b->hist_copy.assign(in->hist, in->hist + TinyBatch::HIST_BINS); else b->hist_copy.clear();
    b->auto_since = b->tile_since = b->regroup_since = 0;
    b->hist_pending = false; b->ls_pending = false; b->probe_was_tile = false; b->probe_was_growth = false; b->auto_last_cap = 0;
    // what the first launch of an imported form would otherwise allocate inside its solve call
    if (b->auto_verdict == 1 && b->auto_cap > 0) {
        if (int rc = ensure_repack_buffers(b)) return rc;
        if (b->repack_sort != 0) { if (int rc = ensure_regroup_buffers(b, false)) return rc; }
    }
    if (b->regroup_verdict == 1) { if (int rc = ensure_regroup_buffers(b, true)) return rc; }
    return TINY_OK;
}

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
