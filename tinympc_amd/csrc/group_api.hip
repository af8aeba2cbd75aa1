// group_api.hip -- the native multi-GPU surface of libtinympc_amd.so (include/tinympc_amd.h, section C).
//
// The reference's callers are single-process C++ programs (examples/quadrotor_hovering.cpp:47-93): a TinyGroup lets one
// such process drive every GPU of the node.  The batch is sharded over the devices -- contiguous blocks, or round-robin
// by instance index when iteration counts diverge -- and each shard is an ordinary TinyBatch on its own
// device and stream: a group solve enqueues every shard's launch without waiting, so the GPUs run concurrently.  There is
// NO data-path collective.  The one exchange is a 64-byte message per shard -- {sum iter, sum solved, accumulated
// iterations, accumulated solves, four residual maxima} -- moved by ONE RCCL all-gather over xGMI (ncclAllGather inside
// ncclGroupStart/End, one communicator per device from ncclCommInitAll) and reduced on the host (SUM / MAX).
// tiny_batch_allreduce_stats is the same exchange for the one-process-per-GPU layout, on a communicator the caller owns.
//
// librccl is dlopen'ed on first use: single-GPU users of the library never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>       // types and prototypes only: the entry points are resolved with dlsym

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "batch_impl.hpp"

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    std::string why;
};

Rccl load_rccl() {
    Rccl r;
    // RCCL must sit on the SAME HIP / HSA runtime this library is bound to.  A process can hold two ROCm stacks -- the
    // system's under /opt/rocm and the one PyTorch-ROCm bundles in torch/lib, whichever got loaded first serves the HIP
    // calls of this library -- and an RCCL from the other stack finds its own, uninitialised HSA runtime ("no ROCm-capable
    // device").  So: the librccl that lives next to the libamdhip64 our HIP symbols resolve to; a bare soname lookup only
    // as the last resort.
    Dl_info hip_lib;
    std::string dir;
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &hip_lib) && hip_lib.dli_fname) {
        dir = hip_lib.dli_fname;
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    if (!dir.empty())
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.so = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
    if (!r.so)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
    if (!r.so) {
        const char* e = dlerror();                   // (reading it clears it: once)
        r.why = e ? e : "librccl.so not found";
        return r;
    }
    auto sym = [&](const char* n) { void* p = dlsym(r.so, n); if (!p) r.why = std::string("missing symbol ") + n; return p; };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        dlclose(r.so);
        r.so = nullptr;
    }
    return r;
}

Rccl& rccl_state() {
    static Rccl r = load_rccl();                     // initialised once, also when several rank threads arrive together
    return r;
}
Rccl* rccl() { return rccl_state().so ? &rccl_state() : nullptr; }
const char* rccl_why() { return rccl_state().why.c_str(); }     // why rccl() is null

enum : int { WIRE = 8 };       // doubles per shard on the wire: 64 bytes

// stats[10] (tiny_batch_reduce_stats layout) -> wire[8] = {sum iter, sum solved, acc iters, acc solved, 4 residual maxima}
__global__ void pack_wire_kernel(const double* __restrict__ stats, double* __restrict__ wire) {
    const int t = threadIdx.x;
    const int src[WIRE] = {0, 1, 7, 8, 3, 4, 5, 6};
    if (t < WIRE) wire[t] = stats[src[t]];
}

void reduce_wire_table(const double* table, int n, double total_batch, double* out10) {
    for (int i = 0; i < 10; ++i) out10[i] = 0.0;
    for (int r = 0; r < n; ++r) {
        const double* w = table + (size_t)r * WIRE;
        out10[0] += w[0]; out10[1] += w[1]; out10[7] += w[2]; out10[8] += w[3];
        // a NaN residual (a diverged shard) must survive the reduction, as it does in the shard-level atomicMax on bit
        // patterns and in distributed.reduce_table (torch.max): std::max(out, NaN) would silently drop it
        for (int k = 0; k < 4; ++k) out10[3 + k] = (std::isnan(w[4 + k]) || std::isnan(out10[3 + k])) ? std::nan("") : std::max(out10[3 + k], w[4 + k]);
    }
    out10[2] = total_batch;
}

size_t field_doubles(int nx, int nu, int N, TinyField f) {
    if (f == TINY_F_X0) return (size_t)nx;
    const bool st = (f == TINY_F_XREF || f == TINY_F_X || f == TINY_F_VNEW || f == TINY_F_G || f == TINY_F_V ||
                     f == TINY_F_VCNEW || f == TINY_F_GC || f == TINY_F_Q || f == TINY_F_P || f == TINY_F_VLNEW ||
                     f == TINY_F_GL || f == TINY_F_VLNEW_TV || f == TINY_F_GL_TV);
    return st ? (size_t)nx * N : (size_t)nu * (N - 1);
}

}  // namespace

struct TinyGroup {
    int nx = 0, nu = 0, N = 0, batch = 0, n = 0;
    bool interleaved = false, use_rccl = false;
    std::vector<TinyBatch*> shard;
    std::vector<int> device, count;
    std::vector<ncclComm_t> comm;
    std::vector<double*> d_stats, d_wire;        // per shard: 10-double statistics, n x 8 gather table (device)
    double* h_table = nullptr;                   // pinned: the gathered table as shard 0 holds it (RCCL) / one row per shard (host mode)
    std::vector<double> stage;
    char err[320] = {0};
};

namespace {

thread_local char g_setup_err[320] = "";      // why the last tiny_group_setup on this thread failed (there is no handle to ask)

int gfail(TinyGroup* g, int code, const char* fmt, ...) {
    if (g) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(g->err, sizeof(g->err), fmt, ap);
        va_end(ap);
    }
    return code;
}
// shard k's instance ids in the caller's order
inline long first_of(const TinyGroup* g, int k) {
    if (g->interleaved) return k;
    long lo = 0;
    for (int r = 0; r < k; ++r) lo += g->count[r];
    return lo;
}
inline long global_index(const TinyGroup* g, int k, long local) { return g->interleaved ? (long)k + local * g->n : first_of(g, k) + local; }

template <class F>
int for_shards(TinyGroup* g, F&& f) {
    for (int k = 0; k < g->n; ++k)
        if (int rc = f(g->shard[k])) return gfail(g, rc, "shard %d (device %d): %s", k, g->device[k], tiny_batch_last_error(g->shard[k]));
    return TINY_OK;
}

}  // namespace

extern "C" {

int tiny_group_setup(TinyGroup** out, const double* Adyn, const double* Bdyn, const double* fdyn, const double* Qdiag,
                     const double* Rdiag, double rho, int nx, int nu, int N, int batch, const int* devices, int n_shards,
                     int interleaved, int verbose) {
    if (!out || !Adyn || !Bdyn || !Qdiag || !Rdiag) return TINY_ERR_NULL;
    *out = nullptr;
    int ndev = tiny_batch_device_count();
    if (ndev <= 0) return TINY_ERR_NO_DEVICE;
    if (n_shards <= 0) n_shards = ndev;                               // default: every GPU of the node
    if (batch < n_shards) return TINY_ERR_DIM;
    TinyGroup* g = new TinyGroup();
    g->nx = nx; g->nu = nu; g->N = N; g->batch = batch; g->n = n_shards; g->interleaved = interleaved != 0;
    std::set<int> distinct;
    for (int k = 0; k < n_shards; ++k) {
        const int d = devices ? devices[k] : k % ndev;
        if (d < 0 || d >= ndev) { delete g; return TINY_ERR_ARG; }
        g->device.push_back(d);
        distinct.insert(d);
        const int base = batch / n_shards, rem = batch % n_shards;    // both splits give the first `rem` shards one more
        g->count.push_back(base + (k < rem ? 1 : 0));
    }
    g_setup_err[0] = 0;
    auto bail = [&](int rc, const char* what, const char* detail) {
        snprintf(g_setup_err, sizeof(g_setup_err), "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
        if (verbose) fprintf(stderr, "tinympc_amd: tiny_group_setup: %s\n", g_setup_err);
        tiny_group_destroy(g);
        return rc;
    };
    for (int k = 0; k < n_shards; ++k) {
        TinyBatch* b = nullptr;
        const int rc = tiny_batch_setup(&b, Adyn, Bdyn, fdyn, Qdiag, Rdiag, rho, nx, nu, N, g->count[k], g->device[k], verbose);
        if (rc) return bail(rc, "tiny_batch_setup of a shard failed", nullptr);
        g->shard.push_back(b);
        double *ds = nullptr, *dw = nullptr;
        if (hipSetDevice(g->device[k]) != hipSuccess || hipMalloc(&ds, 10 * sizeof(double)) != hipSuccess ||
            hipMalloc(&dw, (size_t)n_shards * WIRE * sizeof(double)) != hipSuccess) return bail(TINY_ERR_HIP, "hipMalloc of the statistics buffers", nullptr);
        g->d_stats.push_back(ds); g->d_wire.push_back(dw);
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&g->h_table), (size_t)n_shards * WIRE * sizeof(double), hipHostMallocDefault) != hipSuccess)
        return bail(TINY_ERR_HIP, "hipHostMalloc of the gather table", nullptr);
    // RCCL refuses two ranks on one device: shards that share a GPU (more shards than GPUs -- the single-GPU tests of the
    // sharding logic) exchange their 64-byte messages through host memory instead, with the identical reduction
    g->use_rccl = (int)distinct.size() == n_shards && !getenv("TINYMPC_GROUP_HOST_EXCHANGE");
    if (g->use_rccl) {
        Rccl* r = rccl();
        if (!r) return bail(TINY_ERR_NO_DEVICE, "RCCL unavailable", rccl_why());
        g->comm.assign(n_shards, nullptr);
        for (int k = 0; k < n_shards; ++k) { hipSetDevice(g->device[k]); hipDeviceSynchronize(); }
        (void)hipGetLastError();      // RCCL's own HIP checks would trip over an error some earlier, unrelated call left behind
        const ncclResult_t rc = r->CommInitAll(g->comm.data(), n_shards, g->device.data());
        if (rc != ncclSuccess) {
            g->comm.clear();
            return bail(TINY_ERR_HIP, "ncclCommInitAll", r->GetErrorString(rc));
        }
    }
    *out = g;
    return TINY_OK;
}

int tiny_group_destroy(TinyGroup* g) {
    if (!g) return TINY_ERR_NULL;
    if (!g->comm.empty())
        if (Rccl* r = rccl())
            for (ncclComm_t c : g->comm) if (c) r->CommDestroy(c);
    for (size_t k = 0; k < g->shard.size(); ++k) {
        hipSetDevice(g->device[k]);
        if (k < g->d_stats.size() && g->d_stats[k]) hipFree(g->d_stats[k]);
        if (k < g->d_wire.size() && g->d_wire[k]) hipFree(g->d_wire[k]);
        tiny_batch_destroy(g->shard[k]);
    }
    if (g->h_table) hipHostFree(g->h_table);
    delete g;
    return TINY_OK;
}

int tiny_group_shards(TinyGroup* g) { return g ? g->n : TINY_ERR_NULL; }
TinyBatch* tiny_group_shard(TinyGroup* g, int k) { return (g && k >= 0 && k < g->n) ? g->shard[k] : nullptr; }
int tiny_group_uses_rccl(TinyGroup* g) { return g ? (g->use_rccl ? 1 : 0) : TINY_ERR_NULL; }
const char* tiny_group_last_error(TinyGroup* g) { return g ? g->err : g_setup_err; }   /* NULL: the last failed tiny_group_setup */

int tiny_group_shard_indices(TinyGroup* g, int k, int* idx, int capacity) {
    if (!g || k < 0 || k >= g->n) return TINY_ERR_NULL;
    if (idx)
        for (int i = 0; i < g->count[k] && i < capacity; ++i) idx[i] = (int)global_index(g, k, i);
    return g->count[k];
}

int tiny_group_set_bound_constraints(TinyGroup* g, const double* x_min, const double* x_max, const double* u_min, const double* u_max) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_set_bound_constraints(b, x_min, x_max, u_min, u_max); });
}
int tiny_group_set_cone_constraints(TinyGroup* g, int nsc, const int* Acx, const int* qcx, const double* cx, int nic,
                                    const int* Acu, const int* qcu, const double* cu) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_set_cone_constraints(b, nsc, Acx, qcx, cx, nic, Acu, qcu, cu); });
}
int tiny_group_set_linear_constraints(TinyGroup* g, int ns, const double* Ax, const double* bx, int ni, const double* Au, const double* bu) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_set_linear_constraints(b, ns, Ax, bx, ni, Au, bu); });
}
int tiny_group_set_tv_linear_constraints(TinyGroup* g, int ns, const double* Ax, const double* bx, int ni, const double* Au, const double* bu) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_set_tv_linear_constraints(b, ns, Ax, bx, ni, Au, bu); });
}
int tiny_group_update_settings(TinyGroup* g, double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                               int en_state_bound, int en_input_bound, int en_state_soc, int en_input_soc, int en_state_linear,
                               int en_input_linear, int en_tv_state_linear, int en_tv_input_linear) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) {
        return tiny_batch_update_settings(b, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound, en_input_bound,
                                          en_state_soc, en_input_soc, en_state_linear, en_input_linear, en_tv_state_linear, en_tv_input_linear);
    });
}
int tiny_group_set_option(TinyGroup* g, const char* name, long value) {
    if (!g || !name) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_set_option(b, name, value); });
}

// Per-instance data with the FULL batch axis in host memory ([batch] matrices back to back, the caller's instance order), or
// ONE matrix with TINY_BROADCAST.  Contiguous shards copy straight from the caller's array; round-robin shards go through a
// host staging gather.
int tiny_group_set(TinyGroup* g, TinyField field, const double* src, int flags) {
    if (!g || !src) return TINY_ERR_NULL;
    if (flags & TINY_DEVICE) return gfail(g, TINY_ERR_ARG, "tiny_group_set takes host memory (a device pointer belongs to one shard: use tiny_group_shard)");
    if (flags & TINY_BROADCAST) return for_shards(g, [&](TinyBatch* b) { return tiny_batch_set(b, field, src, TINY_HOST | TINY_BROADCAST); });
    const size_t per = field_doubles(g->nx, g->nu, g->N, field);
    for (int k = 0; k < g->n; ++k) {
        const double* from = src + (size_t)first_of(g, k) * per;
        if (g->interleaved) {
            g->stage.resize((size_t)g->count[k] * per);
            for (long i = 0; i < g->count[k]; ++i) memcpy(g->stage.data() + (size_t)i * per, src + (size_t)global_index(g, k, i) * per, per * sizeof(double));
            from = g->stage.data();
        }
        if (int rc = tiny_batch_set(g->shard[k], field, from, TINY_HOST))
            return gfail(g, rc, "shard %d: %s", k, tiny_batch_last_error(g->shard[k]));
    }
    return TINY_OK;
}

int tiny_group_get(TinyGroup* g, TinyField field, double* dst) {
    if (!g || !dst) return TINY_ERR_NULL;
    const size_t per = field_doubles(g->nx, g->nu, g->N, field);
    for (int k = 0; k < g->n; ++k) {
        double* to = dst + (size_t)first_of(g, k) * per;
        if (g->interleaved) { g->stage.resize((size_t)g->count[k] * per); to = g->stage.data(); }
        if (int rc = tiny_batch_get(g->shard[k], field, to, TINY_HOST))
            return gfail(g, rc, "shard %d: %s", k, tiny_batch_last_error(g->shard[k]));
        if (g->interleaved)
            for (long i = 0; i < g->count[k]; ++i) memcpy(dst + (size_t)global_index(g, k, i) * per, g->stage.data() + (size_t)i * per, per * sizeof(double));
    }
    return TINY_OK;
}

int tiny_group_reset(TinyGroup* g) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_reset(b); });
}

// every shard's launch is enqueued on its own device and stream; nothing waits
int tiny_group_solve_async(TinyGroup* g) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_solve_async(b); });
}

int tiny_group_synchronize(TinyGroup* g) {
    if (!g) return TINY_ERR_NULL;
    return for_shards(g, [&](TinyBatch* b) { return tiny_batch_synchronize(b); });
}

// The one exchange of the path.  Per shard, on its stream behind the solve: statistics reduction -> 64-byte wire message
// at this shard's slot of its gather table -> (RCCL) one in-place ncclAllGather for all shards inside a group call ->
// the table of shard 0 comes back to the host and is reduced there.  out10: the tiny_batch_reduce_stats layout, job-wide.
int tiny_group_allreduce_stats(TinyGroup* g, double* out10) {
    if (!g || !out10) return TINY_ERR_NULL;
    for (int k = 0; k < g->n; ++k) {
        TinyBatch* b = g->shard[k];
        if (int rc = tiny_batch_reduce_stats(b, nullptr, g->d_stats[k])) return gfail(g, rc, "shard %d: %s", k, tiny_batch_last_error(b));
        if (hipSetDevice(g->device[k]) != hipSuccess) return gfail(g, TINY_ERR_HIP, "hipSetDevice(%d)", g->device[k]);   // (a launch goes to the current device's streams only)
        hipLaunchKernelGGL(pack_wire_kernel, dim3(1), dim3(64), 0, b->stream, g->d_stats[k], g->d_wire[k] + (size_t)k * WIRE);
        if (hipGetLastError() != hipSuccess) return gfail(g, TINY_ERR_HIP, "pack_wire_kernel launch failed on shard %d", k);
    }
    if (g->use_rccl) {
        Rccl* r = rccl();
        ncclResult_t rc = r->GroupStart();
        for (int k = 0; k < g->n && rc == ncclSuccess; ++k)
            rc = r->AllGather(g->d_wire[k] + (size_t)k * WIRE, g->d_wire[k], WIRE, ncclDouble, g->comm[k], g->shard[k]->stream);
        const ncclResult_t rc2 = r->GroupEnd();
        if (rc != ncclSuccess || rc2 != ncclSuccess)
            return gfail(g, TINY_ERR_HIP, "ncclAllGather: %s", r->GetErrorString(rc != ncclSuccess ? rc : rc2));
        if (hipSetDevice(g->device[0]) != hipSuccess ||
            hipMemcpyAsync(g->h_table, g->d_wire[0], (size_t)g->n * WIRE * sizeof(double), hipMemcpyDeviceToHost, g->shard[0]->stream) != hipSuccess)
            return gfail(g, TINY_ERR_HIP, "copy of the gathered table failed");
    } else {
        for (int k = 0; k < g->n; ++k)
            if (hipSetDevice(g->device[k]) != hipSuccess ||
                hipMemcpyAsync(g->h_table + (size_t)k * WIRE, g->d_wire[k] + (size_t)k * WIRE, WIRE * sizeof(double), hipMemcpyDeviceToHost, g->shard[k]->stream) != hipSuccess)
                return gfail(g, TINY_ERR_HIP, "copy of shard %d's message failed", k);
    }
    if (int rc = tiny_group_synchronize(g)) return rc;               // every shard done: solve, reduction, exchange
    reduce_wire_table(g->h_table, g->n, (double)g->batch, out10);
    return TINY_OK;
}

// == tiny_solve over the whole group: 0 when every instance on every GPU converged, 1 otherwise (admm.cpp:441,454)
int tiny_group_solve(TinyGroup* g) {
    if (!g) return TINY_ERR_NULL;
    if (int rc = tiny_group_solve_async(g)) return rc;
    double st[10];
    if (int rc = tiny_group_allreduce_stats(g, st)) return rc;
    return st[1] == (double)g->batch ? 0 : 1;
}

int tiny_group_get_status(TinyGroup* g, int* iter, int* solved, int* status, double* residuals) {
    if (!g) return TINY_ERR_NULL;
    std::vector<int> it, so, st;
    std::vector<double> rs;
    for (int k = 0; k < g->n; ++k) {
        const int c = g->count[k];
        it.resize(c); so.resize(c); st.resize(c); rs.resize((size_t)c * 4);
        if (int rc = tiny_batch_get_status(g->shard[k], it.data(), so.data(), st.data(), rs.data()))
            return gfail(g, rc, "shard %d: %s", k, tiny_batch_last_error(g->shard[k]));
        for (long i = 0; i < c; ++i) {
            const long gi = global_index(g, k, i);
            if (iter) iter[gi] = it[i];
            if (solved) solved[gi] = so[i];
            if (status) status[gi] = st[i];
            if (residuals) memcpy(residuals + gi * 4, rs.data() + i * 4, 4 * sizeof(double));
        }
    }
    return TINY_OK;
}

// Communicator plumbing for hosts without an RCCL binding of their own (the ctypes mirror, bench.py): rank 0 draws a
// 128-byte ncclUniqueId, the host distributes it however it talks to its ranks, every rank joins with it.
int tiny_rccl_unique_id(void* id128) {
    if (!id128) return TINY_ERR_NULL;
    Rccl* r = rccl();
    if (!r || !r->GetUniqueId) return TINY_ERR_NO_DEVICE;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return r->GetUniqueId(static_cast<ncclUniqueId*>(id128)) == ncclSuccess ? TINY_OK : TINY_ERR_HIP;
}
int tiny_rccl_comm_init_rank(void** comm, int n_ranks, const void* id128, int rank, int device) {
    if (!comm || !id128) return TINY_ERR_NULL;
    Rccl* r = rccl();
    if (!r || !r->CommInitRank) return TINY_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TINY_ERR_NO_DEVICE;
    (void)hipGetLastError();
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (r->CommInitRank(&c, n_ranks, id, rank) != ncclSuccess) return TINY_ERR_HIP;
    *comm = c;
    return TINY_OK;
}
// The host reduction of the gathered 64-byte messages ({sum iter, sum solved, accumulated iterations, accumulated solves, four
// residual maxima} per shard) -> the 10-entry statistics vector: what tiny_group_allreduce_stats / tiny_batch_allreduce_stats run
// after their all-gather, for hosts that move the messages themselves (MPI, torch.distributed).  No GPU involved.
int tiny_reduce_stats_messages(const double* table, int n_shards, long total_batch, double* out10) {
    if (!table || !out10 || n_shards <= 0) return TINY_ERR_NULL;
    reduce_wire_table(table, n_shards, (double)total_batch, out10);
    return TINY_OK;
}
int tiny_rccl_available(void) {
    Rccl* r = rccl();
    return (r && r->GetUniqueId && r->CommInitRank) ? 1 : 0;
}
// number of ranks of a communicator (ncclCommCount), or a (negative) TINY_ERR_* code: lets a host report the size the
// exchange REALLY runs at instead of the one it asked for
int tiny_rccl_comm_count(void* comm) {
    Rccl* r = rccl();
    if (!r || !r->CommCount || !comm) return TINY_ERR_NULL;
    int n = 0;
    return r->CommCount(static_cast<ncclComm_t>(comm), &n) == ncclSuccess ? n : TINY_ERR_HIP;
}
int tiny_rccl_comm_destroy(void* comm) {
    Rccl* r = rccl();
    if (!r || !comm) return TINY_ERR_NULL;
    return r->CommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? TINY_OK : TINY_ERR_HIP;
}

// The 64-byte wire message of this batch -- {sum iter, sum solved, accumulated iterations, accumulated solves, four
// residual maxima} -- written to device_out (8 doubles of device memory) on the batch's stream, behind the solve: for
// hosts that run the exchange themselves (bench.py hands it to torch.distributed, i.e. RCCL).
int tiny_batch_stats_message(TinyBatch* b, void* device_out) {
    if (!b || !device_out) return TINY_ERR_NULL;
    if (int rc = tiny_batch_reduce_stats(b, nullptr, nullptr)) return rc;          // into the batch's own d_stats
    hipLaunchKernelGGL(pack_wire_kernel, dim3(1), dim3(64), 0, b->stream, b->d_stats, static_cast<double*>(device_out));
    if (hipGetLastError() != hipSuccess) return tinympc_amd::fail(b, TINY_ERR_HIP, "pack_wire_kernel launch failed");
    return TINY_OK;
}

// One process per GPU (MPI / torchrun-style hosts): the same 64-byte exchange on a communicator the CALLER created
// (ncclCommInitRank) -- rccl_comm is its ncclComm_t, n_ranks its size.  Enqueued on the batch's stream behind the solve;
// returns after the stream has drained, with the job-wide statistics in out10 on every rank.
int tiny_batch_allreduce_stats(TinyBatch* b, void* rccl_comm, int n_ranks, int rank, long total_batch, double* out10) {
    if (!b || !rccl_comm || !out10) return TINY_ERR_NULL;
    if (n_ranks <= 0 || rank < 0 || rank >= n_ranks) return tinympc_amd::fail(b, TINY_ERR_ARG, "rank %d of %d", rank, n_ranks);
    Rccl* r = rccl();
    if (!r) return tinympc_amd::fail(b, TINY_ERR_NO_DEVICE, "RCCL unavailable: %s", rccl_why());
    if (hipSetDevice(b->device) != hipSuccess) return TINY_ERR_HIP;
    if (!b->d_wire || b->wire_ranks < n_ranks) {
        if (b->d_wire) hipFree(b->d_wire);
        if (b->h_wire) hipHostFree(b->h_wire);
        b->d_wire = nullptr; b->h_wire = nullptr;
        if (hipMalloc(&b->d_wire, (size_t)n_ranks * WIRE * sizeof(double)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&b->h_wire), (size_t)n_ranks * WIRE * sizeof(double), hipHostMallocDefault) != hipSuccess)
            return tinympc_amd::fail(b, TINY_ERR_HIP, "wire buffers");
        b->wire_ranks = n_ranks;
    }
    if (int rc = tiny_batch_reduce_stats(b, nullptr, nullptr)) return rc;          // into the batch's own d_stats
    hipLaunchKernelGGL(pack_wire_kernel, dim3(1), dim3(64), 0, b->stream, b->d_stats, b->d_wire + (size_t)rank * WIRE);
    if (hipGetLastError() != hipSuccess) return tinympc_amd::fail(b, TINY_ERR_HIP, "pack_wire_kernel launch failed");
    const ncclResult_t rc = r->AllGather(b->d_wire + (size_t)rank * WIRE, b->d_wire, WIRE, ncclDouble, (ncclComm_t)rccl_comm, b->stream);
    if (rc != ncclSuccess) return tinympc_amd::fail(b, TINY_ERR_HIP, "ncclAllGather: %s", r->GetErrorString(rc));
    if (hipMemcpyAsync(b->h_wire, b->d_wire, (size_t)n_ranks * WIRE * sizeof(double), hipMemcpyDeviceToHost, b->stream) != hipSuccess ||
        hipStreamSynchronize(b->stream) != hipSuccess)
        return tinympc_amd::fail(b, TINY_ERR_HIP, "copy of the gathered table failed");
    reduce_wire_table(b->h_wire, n_ranks, (double)total_batch, out10);
    return TINY_OK;
}

}  // extern "C"
