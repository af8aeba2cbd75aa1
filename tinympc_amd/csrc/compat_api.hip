// compat_api.hip -- the reference's own entry points (include/tinympc_amd.h, part B) over plain-data
// mirrors of TinySolver / TinyCache / TinyWorkspace / TinySettings / TinySolution
// (reference src/tinympc/types.hpp:32-218), so a caller compiled against the reference's headers can
// link this library instead of libtinympcstatic.a.
//
// tiny_solve(TinySolver*) == a batch of ONE on the GPU: the workspace is gathered into the
// device records, the same admm_solve_kernel runs, and every workspace field the reference's
// solve() touches (x,u,q,r,p,d,v,vnew,z,znew,g,y,vcnew,zcnew,gc,yc, residuals, status, iter,
// solution) is scattered back.  It is a latency-bound convenience/parity path; throughput users
// call tiny_batch_* directly.  There is no CPU solve in here.
#include "batch_impl.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <chrono>
#include <map>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unistd.h>

using namespace tinympc_amd;

namespace {
#include "_gen/sensitivity.inc"      // k_dKinf_drho[48], k_dPinf_drho[144], k_dC1_drho[16], k_dC2_drho[144], column-major

// ---- Eigen-compatible storage: with default x86-64 flags Eigen allocates with plain malloc/free
// (EIGEN_MALLOC_ALREADY_ALIGNED, Eigen/src/Core/util/Memory.h:23-57), so buffers made here can be
// resized / freed by Eigen-side callers and vice versa.
void mat_alloc(TinyMatrixPOD* m, int64_t r, int64_t c) {
    m->data = (r * c) ? (double*)calloc((size_t)(r * c), sizeof(double)) : nullptr;
    m->rows = r; m->cols = c;
}
void vec_alloc(TinyVectorPOD* v, int64_t r) {
    v->data = r ? (double*)calloc((size_t)r, sizeof(double)) : nullptr;
    v->rows = r;
}
void mat_assign(TinyMatrixPOD* m, const double* src, int64_t r, int64_t c) {   // Eigen operator= semantics
    if (m->rows * m->cols != r * c) {
        free(m->data);
        m->data = (r * c) ? (double*)malloc((size_t)(r * c) * sizeof(double)) : nullptr;
    }
    m->rows = r; m->cols = c;
    if (r * c) memcpy(m->data, src, (size_t)(r * c) * sizeof(double));
}
void vec_assign(TinyVectorPOD* v, const double* src, int64_t r) {
    if (v->rows != r) { free(v->data); v->data = r ? (double*)malloc((size_t)r * sizeof(double)) : nullptr; }
    v->rows = r;
    if (r) memcpy(v->data, src, (size_t)r * sizeof(double));
}
void ivec_assign(TinyVectorXiPOD* v, const int* src, int64_t r) {
    if (v->rows != r) { free(v->data); v->data = r ? (int*)malloc((size_t)r * sizeof(int)) : nullptr; }
    v->rows = r;
    if (r) memcpy(v->data, src, (size_t)r * sizeof(int));
}
void mat_from(TinyMatrixPOD* m, const Mat& s) { mat_assign(m, s.a.data(), s.r, s.c); }

int check_dimension(const char* name, const char* what, long actual, long expected) {   // tiny_api.cpp:13-19
    if (actual != expected) {
        std::cout << name << " has " << actual << " " << what << ". Expected " << expected << "." << std::endl;
        return 1;
    }
    return 0;
}

// Eigen's operator<< with IOFormat(4, 0, ", ", "\n", "[", "]") (tiny_api.cpp:11): every coefficient
// printed at precision 4, right-aligned to the widest one, rows wrapped in [ ].
std::string fmt_matrix(const double* d, int64_t rows, int64_t cols) {
    std::vector<std::string> cell((size_t)(rows * cols));
    size_t width = 0;
    for (int64_t j = 0; j < cols; ++j)
        for (int64_t i = 0; i < rows; ++i) {
            std::ostringstream os;
            os.precision(4);
            os << d[j * rows + i];
            cell[(size_t)(j * rows + i)] = os.str();
            width = std::max(width, os.str().size());
        }
    std::ostringstream out;
    for (int64_t i = 0; i < rows; ++i) {
        if (i) out << "\n";
        out << "[";
        for (int64_t j = 0; j < cols; ++j) {
            if (j) out << ", ";
            out << std::setw((int)width) << cell[(size_t)(j * rows + i)];
        }
        out << "]";
    }
    return out.str();
}
std::string fmt(const Mat& m) { return fmt_matrix(m.a.data(), m.r, m.c); }

Mat to_mat(const TinyMatrixPOD* m) { return Mat((int)m->rows, (int)m->cols, m->data); }

// ---- device contexts for the struct-level solve --------------------------------------------
struct Ctx {
    TinyBatch* b = nullptr;
    int n = 0;
    int nx = 0, nu = 0, N = 0;     // the shape the batch was created for: a freed solver's address can come back with another shape
    // the problem family last uploaded (hash of cache, dynamics, costs, settings, bounds, cones, half-spaces): a solve
    // whose family is unchanged skips the table rebuild and upload
    uint64_t family = 0;
    bool family_valid = false;
    // one pinned host buffer + one device buffer carry every workspace field of every solver in ONE copy per direction
    double* h_pin = nullptr;
    double* d_xfer = nullptr;
    size_t xfer_doubles = 0;
    hipEvent_t ev[8] = {};         // download groups of a large solve_group: the scatter of a group starts when ITS copy has landed
    // Round 6: a context is locked by the call that uses it, not the process.  The reference's tiny_solve on DISTINCT solvers is
    // re-entrant (its only global is the print format, tiny_api.cpp:11); so is this one: every context has its own TinyBatch, stream,
    // staging buffers -- and this mutex.  Two threads on the SAME solver serialise here (the reference would race).
    std::mutex mu;
};
void release(Ctx& c) {             // (c.mu held, or the context unreachable)
    for (hipEvent_t& e : c.ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (c.b) tiny_batch_destroy(c.b);
    if (c.h_pin) hipHostFree(c.h_pin);
    if (c.d_xfer) hipFree(c.d_xfer);
    c.b = nullptr; c.n = c.nx = c.nu = c.N = 0; c.family = 0; c.family_valid = false; c.h_pin = nullptr; c.d_xfer = nullptr; c.xfer_doubles = 0;
}
// the map of contexts is all the process-wide lock guards: look-up, insertion, removal (microseconds)
std::map<TinySolver*, std::shared_ptr<Ctx>> g_ctx;
std::mutex g_mu;
std::shared_ptr<Ctx> context_of(TinySolver* s) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::shared_ptr<Ctx>& p = g_ctx[s];
    if (!p) p = std::make_shared<Ctx>();
    return p;
}
void forget_context(TinySolver* s, const std::shared_ptr<Ctx>& c) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(s);
    if (it != g_ctx.end() && it->second == c) g_ctx.erase(it);
}

int raw_batch(TinyBatch** out, int nx, int nu, int N, int n) {
    // a TinyBatch whose cache is filled from the caller's TinyCache at every solve
    std::vector<double> A((size_t)nx * nx, 0.0), B((size_t)nx * nu, 0.0), Q(nx, 1.0), R(nu, 1.0);
    for (int i = 0; i < nx; ++i) A[(size_t)i * nx + i] = 0.5;
    return tiny_batch_setup(out, A.data(), B.data(), nullptr, Q.data(), R.data(), 1.0, nx, nu, N, n, 0, 0);
}

bool shaped(const TinyMatrixPOD& m, int64_t r, int64_t c) { return m.data && m.rows == r && m.cols == c; }

// copy the problem family (cache, dynamics, costs, bounds, cones, settings) of `s` into the batch
int sync_family(TinyBatch* b, const TinySolver* s) {
    const TinyWorkspace* w = s->work;
    const TinyCache* c = s->cache;
    const TinySettings* st = s->settings;
    const int nx = w->nx, nu = w->nu, N = w->N;
    if (st->adaptive_rho) {                          // admm.cpp:397-423: needs the d(.)/d(rho) tables in the cache
        if (!shaped(c->dKinf_drho, nu, nx) || !shaped(c->dPinf_drho, nx, nx))
            return fail(b, TINY_ERR_DIM, "adaptive_rho is set but cache->dKinf_drho / dPinf_drho are not %d x %d / %d x %d "
                                         "(tiny_initialize_sensitivity_matrices)", nu, nx, nx, nx);
        if (int rc = tiny_batch_set_sensitivity(b, c->dKinf_drho.data, c->dPinf_drho.data, shaped(c->dC1_drho, nu, nu) ? c->dC1_drho.data : nullptr,
                                                shaped(c->dC2_drho, nx, nx) ? c->dC2_drho.data : nullptr)) return rc;
    }
    tiny_batch_set_adaptive_rho(b, st->adaptive_rho, st->adaptive_rho_min, st->adaptive_rho_max, st->adaptive_rho_enable_clipping);
    if (!shaped(c->Kinf, nu, nx) || !shaped(c->Pinf, nx, nx) || !shaped(c->Quu_inv, nu, nu) || !shaped(c->AmBKt, nx, nx) ||
        c->APf.rows != nx || c->BPf.rows != nu || !shaped(w->Adyn, nx, nx) || !shaped(w->Bdyn, nx, nu))
        return fail(b, TINY_ERR_DIM, "cache / dynamics have unexpected shapes");
    b->cache.rho = c->rho;
    b->cache.Kinf = to_mat(&c->Kinf); b->cache.Pinf = to_mat(&c->Pinf);
    b->cache.Quu_inv = to_mat(&c->Quu_inv); b->cache.AmBKt = to_mat(&c->AmBKt);
    b->cache.APf = Mat(nx, 1, c->APf.data); b->cache.BPf = Mat(nu, 1, c->BPf.data);
    b->A = to_mat(&w->Adyn); b->B = to_mat(&w->Bdyn);
    b->f = (w->fdyn.rows == nx) ? Mat(nx, 1, w->fdyn.data) : Mat(nx, 1);
    b->Qw.assign(w->Q.data, w->Q.data + nx); b->Rw.assign(w->R.data, w->R.data + nu);
    b->set.abs_pri_tol = st->abs_pri_tol; b->set.abs_dua_tol = st->abs_dua_tol; b->set.max_iter = st->max_iter;
    b->set.check_termination = st->check_termination; b->set.en_state_bound = st->en_state_bound;
    b->set.en_input_bound = st->en_input_bound; b->set.en_state_soc = st->en_state_soc; b->set.en_input_soc = st->en_input_soc;
    b->set.en_state_linear = st->en_state_linear; b->set.en_input_linear = st->en_input_linear;
    b->set.en_tv_state_linear = st->en_tv_state_linear; b->set.en_tv_input_linear = st->en_tv_input_linear;
    if (st->en_state_linear || st->en_input_linear) {
        if (w->Alin_x.rows != w->numStateLinear || w->Alin_u.rows != w->numInputLinear ||
            (w->numStateLinear && w->Alin_x.cols != nx) || (w->numInputLinear && w->Alin_u.cols != nu))
            return fail(b, TINY_ERR_DIM, "linear constraints enabled but not set (tiny_set_linear_constraints)");
        if (int rc = tiny_batch_set_linear_constraints(b, w->numStateLinear, w->Alin_x.data, w->blin_x.data, w->numInputLinear,
                                                       w->Alin_u.data, w->blin_u.data)) return rc;
    }
    if (st->en_tv_state_linear || st->en_tv_input_linear) {
        if (w->tv_Alin_x.rows != (int64_t)w->numtvStateLinear * N || w->tv_Alin_u.rows != (int64_t)w->numtvInputLinear * (N - 1))
            return fail(b, TINY_ERR_DIM, "time-varying linear constraints enabled but not set (tiny_set_tv_linear_constraints)");
        if (int rc = tiny_batch_set_tv_linear_constraints(b, w->numtvStateLinear, w->tv_Alin_x.data, w->tv_blin_x.data,
                                                          w->numtvInputLinear, w->tv_Alin_u.data, w->tv_blin_u.data)) return rc;
    }
    b->have_bounds = false;
    if (st->en_state_bound || st->en_input_bound) {
        if (!shaped(w->x_min, nx, N) || !shaped(w->x_max, nx, N) || !shaped(w->u_min, nu, N - 1) || !shaped(w->u_max, nu, N - 1))
            return fail(b, TINY_ERR_DIM, "bounds enabled but x_min/x_max/u_min/u_max are not set (tiny_set_bound_constraints)");
        b->x_min.assign(w->x_min.data, w->x_min.data + (size_t)nx * N);
        b->x_max.assign(w->x_max.data, w->x_max.data + (size_t)nx * N);
        b->u_min.assign(w->u_min.data, w->u_min.data + (size_t)nu * (N - 1));
        b->u_max.assign(w->u_max.data, w->u_max.data + (size_t)nu * (N - 1));
        b->have_bounds = true;
    }
    b->tab_dirty = true;
    // a family whose cone switch is off never reaches the cone code of the reference (admm.cpp:102,107): its descriptors are
    // not validated or forwarded, so that e.g. q != 3 cones with en_*_soc = 0 solve as they do there
    const int nsc = st->en_state_soc ? w->numStateCones : 0, nic = st->en_input_soc ? w->numInputCones : 0;
    return tiny_batch_set_cone_constraints(b, nsc, w->Acx.data, w->qcx.data, w->cx.data, nic, w->Acu.data, w->qcu.data, w->cu.data);
}

struct FieldMap { TinyField f; TinyMatrixPOD TinyWorkspace::*m; };
const FieldMap kIn[] = {{TINY_F_XREF, &TinyWorkspace::Xref}, {TINY_F_UREF, &TinyWorkspace::Uref},
                        {TINY_F_VNEW, &TinyWorkspace::vnew}, {TINY_F_ZNEW, &TinyWorkspace::znew},
                        {TINY_F_G, &TinyWorkspace::g}, {TINY_F_Y, &TinyWorkspace::y},
                        {TINY_F_V, &TinyWorkspace::v}, {TINY_F_Z, &TinyWorkspace::z}};
const FieldMap kInSoc[] = {{TINY_F_X, &TinyWorkspace::x}, {TINY_F_U, &TinyWorkspace::u},
                           {TINY_F_GC, &TinyWorkspace::gc}, {TINY_F_YC, &TinyWorkspace::yc}};
const FieldMap kOut[] = {{TINY_F_X, &TinyWorkspace::x}, {TINY_F_U, &TinyWorkspace::u},
                         {TINY_F_VNEW, &TinyWorkspace::vnew}, {TINY_F_ZNEW, &TinyWorkspace::znew},
                         {TINY_F_G, &TinyWorkspace::g}, {TINY_F_Y, &TinyWorkspace::y},
                         {TINY_F_V, &TinyWorkspace::v}, {TINY_F_Z, &TinyWorkspace::z},
                         {TINY_F_Q, &TinyWorkspace::q}, {TINY_F_R, &TinyWorkspace::r},
                         {TINY_F_P, &TinyWorkspace::p}, {TINY_F_D, &TinyWorkspace::d}};
const FieldMap kOutSocS[] = {{TINY_F_VCNEW, &TinyWorkspace::vcnew}, {TINY_F_GC, &TinyWorkspace::gc}};
const FieldMap kOutSocI[] = {{TINY_F_ZCNEW, &TinyWorkspace::zcnew}, {TINY_F_YC, &TinyWorkspace::yc}};
// linear / time-varying linear: duals in, slack + duals out; x,u in (the slack is initialised from them, admm.cpp:361-375)
const FieldMap kInLinS[] = {{TINY_F_GL, &TinyWorkspace::gl}};
const FieldMap kInLinI[] = {{TINY_F_YL, &TinyWorkspace::yl}};
const FieldMap kInTvS[] = {{TINY_F_GL_TV, &TinyWorkspace::gl_tv}};
const FieldMap kInTvI[] = {{TINY_F_YL_TV, &TinyWorkspace::yl_tv}};
const FieldMap kOutLinS[] = {{TINY_F_VLNEW, &TinyWorkspace::vlnew}, {TINY_F_GL, &TinyWorkspace::gl}};
const FieldMap kOutLinI[] = {{TINY_F_ZLNEW, &TinyWorkspace::zlnew}, {TINY_F_YL, &TinyWorkspace::yl}};
const FieldMap kOutTvS[] = {{TINY_F_VLNEW_TV, &TinyWorkspace::vlnew_tv}, {TINY_F_GL_TV, &TinyWorkspace::gl_tv}};
const FieldMap kOutTvI[] = {{TINY_F_ZLNEW_TV, &TinyWorkspace::zlnew_tv}, {TINY_F_YL_TV, &TinyWorkspace::yl_tv}};

bool is_state_field(TinyField f) {
    return f == TINY_F_XREF || f == TINY_F_X || f == TINY_F_VNEW || f == TINY_F_G || f == TINY_F_V || f == TINY_F_VCNEW ||
           f == TINY_F_GC || f == TINY_F_Q || f == TINY_F_P || f == TINY_F_VLNEW || f == TINY_F_GL || f == TINY_F_VLNEW_TV ||
           f == TINY_F_GL_TV;
}

// the device context (a TinyBatch of n instances) that backs solver s0; ctx.mu must be held
int device_context(Ctx& ctx, TinySolver* s0, int n, TinyBatch** out) {
    const int nx = s0->work->nx, nu = s0->work->nu, N = s0->work->N;
    // The reference has no destroy function, so a caller may free a solver and tiny_setup() another one of a different
    // shape at the same address: the context is only reused when group size AND (nx, nu, N) still match.
    if (ctx.b && (ctx.n != n || ctx.nx != nx || ctx.nu != nu || ctx.N != N)) release(ctx);
    if (!ctx.b) {
        int rc = raw_batch(&ctx.b, nx, nu, N, n);
        if (rc) {
            fprintf(stderr, "tiny_solve: cannot create the device context for (nx,nu,N)=(%d,%d,%d): error %d "
                            "(libtinympc_amd has no CPU path)\n", nx, nu, N, rc);
            ctx.b = nullptr;
            return rc;
        }
        ctx.n = n; ctx.nx = nx; ctx.nu = nu; ctx.N = N;
        tiny_batch_set_option(ctx.b, "debug", 1);
    }
    *out = ctx.b;
    return TINY_OK;
}

// host-side gather / scatter over many TinySolver structs is memory-latency bound (thousands of small heap blocks):
// the solver range is split over a few threads when the group is large.  The workers are created once and parked on a
// condition variable -- spawning 16 threads twice per tiny_solve_batch cost more than the copies they made.
class WorkerPool {
  public:
    explicit WorkerPool(int workers) {
        for (int w = 0; w < workers; ++w) th_.emplace_back([this, w] { loop(w + 1); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    // job(t) for t = 0 .. parts-1 (parts <= workers + 1); part 0 runs on the calling thread; returns when all are done
    void run(int parts, const std::function<void(int)>& job) {
        { std::lock_guard<std::mutex> lk(mu_); job_ = &job; parts_ = parts; pending_ = parts - 1; ++gen_; }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(int id) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (id < parts_) job = job_;
            }
            if (job) {
                (*job)(id);
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* job_ = nullptr;
    int parts_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

template <class F>
void parallel_solvers(int n, F&& body) {
    const unsigned hw = std::thread::hardware_concurrency();
    // 16 measured best on 2 x EPYC 9575F (8: 3.4 ms, 16: 2.8 ms, 24 / 32: 3.3 ms per 4 096 quadrotor solvers); at most 32 (per-thread line buffers)
    static const int cap = getenv("TINYMPC_AMD_HOST_THREADS") ? std::min(32, std::max(1, atoi(getenv("TINYMPC_AMD_HOST_THREADS")))) : 16;
    const int nt = (n < 512) ? 1 : (int)std::min<unsigned>((unsigned)cap, std::max(1u, hw / 2));
    if (nt <= 1) { body(0, n, 0); return; }
    // (ONE pool for the process: concurrent large groups take turns at it -- pool_mu.)  The parked threads belong to the process that created them: a fork()ed
    // child has none of them, and waiting on pending_ there would never end -- so the pool is per pid; the parent's pool object
    // is abandoned in the child (its threads do not exist there: joining them would hang as well).
    static std::mutex pool_mu;
    std::lock_guard<std::mutex> pool_lk(pool_mu);
    static WorkerPool* pool_ptr = nullptr;
    static pid_t pool_pid = 0;
    if (!pool_ptr || pool_pid != getpid()) { pool_ptr = new WorkerPool(cap - 1); pool_pid = getpid(); }
    WorkerPool& pool = *pool_ptr;
    const int parts = std::min(nt, pool.workers() + 1);
    const int per = (n + parts - 1) / parts;
    const std::function<void(int)> job = [&](int t) {
        const int lo = t * per, hi = std::min(n, lo + per);
        if (lo < hi) body(lo, hi, t);
    };
    pool.run(parts, job);
}

// (eight bytes per step since round 6: byte by byte the ~6 KB of a quadrotor family cost 6 of a tiny_solve call's 43 us; the value
// is only ever compared with the value the same function gave the call before)
uint64_t fnv(uint64_t h, const void* p, size_t bytes) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        memcpy(&w, c + i, 8);
        h = (h ^ w) * 1099511628211ull;
        h ^= h >> 29;
    }
    for (; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
uint64_t family_hash(const TinySolver* s) {
    const TinyWorkspace* w = s->work;
    const TinyCache* c = s->cache;
    uint64_t h = 1469598103934665603ull;
    auto mat = [&](const TinyMatrixPOD& m) { h = fnv(h, &m.rows, 16); if (m.data) h = fnv(h, m.data, (size_t)(m.rows * m.cols) * 8); };
    auto vec = [&](const TinyVectorPOD& v) { h = fnv(h, &v.rows, 8); if (v.data) h = fnv(h, v.data, (size_t)v.rows * 8); };
    auto ivec = [&](const TinyVectorXiPOD& v) { h = fnv(h, &v.rows, 8); if (v.data) h = fnv(h, v.data, (size_t)v.rows * 4); };
    h = fnv(h, s->settings, sizeof(TinySettings));
    h = fnv(h, &c->rho, 8);
    mat(c->Kinf); mat(c->Pinf); mat(c->Quu_inv); mat(c->AmBKt); vec(c->APf); vec(c->BPf);
    if (s->settings->adaptive_rho) { mat(c->dKinf_drho); mat(c->dPinf_drho); mat(c->dC1_drho); mat(c->dC2_drho); }
    mat(w->Adyn); mat(w->Bdyn); vec(w->fdyn); vec(w->Q); vec(w->R);
    mat(w->x_min); mat(w->x_max); mat(w->u_min); mat(w->u_max);
    h = fnv(h, &w->numStateCones, 8); vec(w->cx); vec(w->cu); ivec(w->Acx); ivec(w->Acu); ivec(w->qcx); ivec(w->qcu);
    h = fnv(h, &w->numStateLinear, 8); mat(w->Alin_x); vec(w->blin_x); mat(w->Alin_u); vec(w->blin_u);
    h = fnv(h, &w->numtvStateLinear, 8); mat(w->tv_Alin_x); mat(w->tv_blin_x); mat(w->tv_Alin_u); mat(w->tv_blin_u);
    return h;
}

int solve_group_locked(Ctx& ctx, TinySolver** solvers, int n, TinyBatch** bout);
// Every error return of the body below may leave asynchronous copies into the context's pinned / transfer buffers in flight
// (uploads before the solve, the event-ordered downloads after it); the next call reuses those buffers, so a failed call
// drains the stream (best effort) before it reports.
int solve_group(TinySolver** solvers, int n) {
    if (!solvers || n <= 0 || !solvers[0]) return TINY_ERR_NULL;
    const std::shared_ptr<Ctx> ctx = context_of(solvers[0]);
    std::lock_guard<std::mutex> lk(ctx->mu);
    TinyBatch* b = nullptr;
    const int rc = solve_group_locked(*ctx, solvers, n, &b);
    if (!ctx->b) forget_context(solvers[0], ctx);       // (no device context could be made: nothing to keep)
    if (rc != TINY_OK && b && b->stream)      // (1 = max_iter reached shares its value with TINY_ERR_DIM: the stream is idle then, the call is free)
        { (void)hipStreamSynchronize(b->stream); (void)hipGetLastError(); }
    return rc;
}
int solve_group_locked(Ctx& ctx, TinySolver** solvers, int n, TinyBatch** bout) {
    TinySolver* s0 = solvers[0];
    const int nx = s0->work->nx, nu = s0->work->nu, N = s0->work->N;
    TinyBatch* b = nullptr;
    if (int rc = device_context(ctx, s0, n, &b)) return rc;
    *bout = b;
    const uint64_t fam = family_hash(s0);
    if (!ctx.family_valid || ctx.family != fam) {
        ctx.family_valid = false;
        if (int rc = sync_family(b, s0)) { fprintf(stderr, "tiny_solve: %s\n", b->err); return rc; }
        ctx.family = fam; ctx.family_valid = true;
    }
    const size_t ns = (size_t)nx * N, ni = (size_t)nu * (N - 1);
    const TinySettings* st0 = s0->settings;
    const bool s_soc = st0->en_state_soc && s0->work->numStateCones > 0;
    const bool i_soc = st0->en_input_soc && s0->work->numInputCones > 0;
    const bool any_lin = st0->en_state_linear || st0->en_input_linear || st0->en_tv_state_linear || st0->en_tv_input_linear;
    // transfer plan: which workspace fields go up, which come back (exactly what the reference's solve() reads / writes)
    std::vector<FieldMap> in(std::begin(kIn), std::end(kIn)), out(std::begin(kOut), std::end(kOut));
    if (s_soc || i_soc || any_lin)
        for (const FieldMap& fm : kInSoc)
            if (!((fm.f == TINY_F_GC || fm.f == TINY_F_YC) && !(s_soc || i_soc))) in.push_back(fm);
    auto add = [](std::vector<FieldMap>& v, const FieldMap* a, size_t k) { v.insert(v.end(), a, a + k); };
    if (st0->en_state_linear) { add(in, kInLinS, 1); add(out, kOutLinS, 2); }
    if (st0->en_input_linear) { add(in, kInLinI, 1); add(out, kOutLinI, 2); }
    if (st0->en_tv_state_linear) { add(in, kInTvS, 1); add(out, kOutTvS, 2); }
    if (st0->en_tv_input_linear) { add(in, kInTvI, 1); add(out, kOutTvI, 2); }
    if (s_soc) add(out, kOutSocS, 2);
    if (i_soc) add(out, kOutSocI, 2);
    auto fsize = [&](const FieldMap& fm) { return is_state_field(fm.f) ? ns : ni; };
    size_t in_doubles = (size_t)n * nx, out_doubles = (size_t)n * 6;      // + x0 | + status (int4 = 2 doubles) + 4 residuals
    for (const FieldMap& fm : in) in_doubles += (size_t)n * fsize(fm);
    for (const FieldMap& fm : out) out_doubles += (size_t)n * fsize(fm);
    const size_t need = std::max(in_doubles, out_doubles);
    if (ctx.xfer_doubles < need) {
        if (ctx.h_pin) hipHostFree(ctx.h_pin);
        if (ctx.d_xfer) hipFree(ctx.d_xfer);
        ctx.h_pin = nullptr; ctx.d_xfer = nullptr; ctx.xfer_doubles = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx.h_pin), need * sizeof(double), hipHostMallocDefault) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx.d_xfer), need * sizeof(double)) != hipSuccess) return TINY_ERR_HIP;
        ctx.xfer_doubles = need;
    }
    // gather -> one H2D copy -> device-side packs (no host synchronisation until the results are back)
    const bool trace = getenv("TINYMPC_AMD_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    for (const FieldMap& fm : in)
        for (int k = 0; k < n; ++k) {
            const TinyMatrixPOD& m = solvers[k]->work->*(fm.m);
            if ((size_t)(m.rows * m.cols) != fsize(fm) || !m.data) return fail(b, TINY_ERR_DIM, "workspace field %d of solver %d has the wrong size", (int)fm.f, k);
        }
    // Large groups move their fields in a few GROUPS of consecutive fields (the staging buffer is field-major, so a group is one
    // contiguous block): while the host threads gather group g+1 the copy engine uploads group g, and on the way back the
    // scatter of group g runs while group g+1 is still crossing PCIe -- the copies (0.4 + 0.6 ms for 4 096 quadrotor solvers)
    // disappear behind the host work that has to happen anyway.
    const int groups = n >= 512 ? 4 : 1;
    // A handful of solvers (tiny_solve = one): the pack / unpack kernels read and write the pinned host buffer directly over
    // PCIe (it is device-accessible) -- two copy commands less in a call that is all latency.
    const bool zero_copy = n <= 8 && !getenv("TINYMPC_AMD_NO_ZERO_COPY");
    double* const xbuf = zero_copy ? ctx.h_pin : ctx.d_xfer;
    auto split_groups = [&](const std::vector<FieldMap>& fs, size_t tail_doubles) {      // -> field index bounds of each group
        size_t total = tail_doubles;
        for (const FieldMap& fm : fs) total += (size_t)n * fsize(fm);
        std::vector<size_t> bound{0};
        size_t acc = 0;
        for (size_t i = 0; i < fs.size(); ++i) {
            acc += (size_t)n * fsize(fs[i]);
            if ((int)bound.size() < groups && acc >= total * bound.size() / groups) bound.push_back(i + 1);
        }
        if (bound.back() != fs.size()) bound.push_back(fs.size());
        return bound;
    };
    {
        const std::vector<size_t> gb = split_groups(in, (size_t)n * nx);
        size_t o0 = 0;
        for (size_t g = 0; g + 1 < gb.size(); ++g) {
            const bool last = g + 2 == gb.size();
            size_t o1 = o0;
            for (size_t i = gb[g]; i < gb[g + 1]; ++i) o1 += (size_t)n * fsize(in[i]);
            parallel_solvers(n, [&](int lo, int hi, int) {
                size_t o = o0;
                for (size_t i = gb[g]; i < gb[g + 1]; ++i) {
                    const FieldMap& fm = in[i];
                    const size_t sz = fsize(fm);
                    for (int k = lo; k < hi; ++k) memcpy(ctx.h_pin + o + k * sz, (solvers[k]->work->*(fm.m)).data, sz * sizeof(double));
                    o += (size_t)n * sz;
                }
                if (last) for (int k = lo; k < hi; ++k) memcpy(ctx.h_pin + o + (size_t)k * nx, solvers[k]->work->x.data, nx * sizeof(double));   // x[:,0] = x0
            });
            if (last) o1 += (size_t)n * nx;
            if (!zero_copy && hipMemcpyAsync(ctx.d_xfer + o0, ctx.h_pin + o0, (o1 - o0) * sizeof(double), hipMemcpyHostToDevice, b->stream) != hipSuccess) return TINY_ERR_HIP;
            o0 = o1;
        }
    }
    size_t off = 0;
    const double t1 = now();
    {   // device-side unpack of every uploaded field (+ x0) into the records: ONE launch
        std::vector<TinyField> fs;
        std::vector<size_t> offs;
        off = 0;
        for (const FieldMap& fm : in) { fs.push_back(fm.f); offs.push_back(off); off += (size_t)n * fsize(fm); }
        fs.push_back(TINY_F_X0); offs.push_back(off);
        if (int rc = xfer_fields(b, fs.data(), offs.data(), (int)fs.size(), xbuf, true, false, 0, 0)) { fprintf(stderr, "tiny_solve: %s\n", b->err); return rc; }
    }
    // adaptive rho: every solver's own cache is state (rho, Kinf, Pinf and the dead copies C1, C2 move during the solve and
    // persist, rho_benchmark.cpp:196-210): up before the launch, back into the caller's TinyCache after it
    struct CachePart { const char* name; size_t per; TinyMatrixPOD TinyCache::*m; };
    const CachePart cparts[] = {{"Kinf", (size_t)nu * nx, &TinyCache::Kinf}, {"Pinf", (size_t)nx * nx, &TinyCache::Pinf},
                                {"C1", (size_t)nu * nu, &TinyCache::C1}, {"C2", (size_t)nx * nx, &TinyCache::C2}};
    const bool adaptive = st0->adaptive_rho != 0;
    std::vector<double> cbuf;
    if (adaptive) {
        cbuf.resize((size_t)n);
        for (int k = 0; k < n; ++k) cbuf[k] = solvers[k]->cache->rho;
        if (int rc = tiny_batch_set_cache_state(b, "rho", cbuf.data())) { fprintf(stderr, "tiny_solve: %s\n", b->err); return rc; }
        for (const CachePart& cp : cparts) {
            cbuf.assign((size_t)n * cp.per, 0.0);
            for (int k = 0; k < n; ++k) {
                const TinyMatrixPOD& m = solvers[k]->cache->*(cp.m);
                if ((size_t)(m.rows * m.cols) != cp.per || !m.data) return fail(b, TINY_ERR_DIM, "cache->%s of solver %d has the wrong size", cp.name, k);
                memcpy(cbuf.data() + (size_t)k * cp.per, m.data, cp.per * sizeof(double));
            }
            if (int rc = tiny_batch_set_cache_state(b, cp.name, cbuf.data())) { fprintf(stderr, "tiny_solve: %s\n", b->err); return rc; }
        }
    }

    if (int rc = launch_solve(b)) { fprintf(stderr, "tiny_solve: %s\n", b->err); return rc; }
    if (adaptive) {
        cbuf.resize((size_t)n);
        if (int rc = tiny_batch_get_cache_state(b, "rho", cbuf.data())) return rc;
        for (int k = 0; k < n; ++k) solvers[k]->cache->rho = cbuf[k];
        for (const CachePart& cp : cparts) {
            cbuf.resize((size_t)n * cp.per);
            if (int rc = tiny_batch_get_cache_state(b, cp.name, cbuf.data())) return rc;
            for (int k = 0; k < n; ++k) memcpy((solvers[k]->cache->*(cp.m)).data, cbuf.data() + (size_t)k * cp.per, cp.per * sizeof(double));
        }
        ctx.family_valid = false;                    // the caches moved: the next solve re-reads the family
    }

    off = 0;
    for (const FieldMap& fm : out) off += (size_t)n * fsize(fm);
    const size_t off_status = off, off_resid = off + (size_t)n * 2;
    {   // every downloaded field + status + residuals into the transfer buffer: ONE launch
        std::vector<TinyField> fs;
        std::vector<size_t> offs;
        size_t o = 0;
        for (const FieldMap& fm : out) { fs.push_back(fm.f); offs.push_back(o); o += (size_t)n * fsize(fm); }
        if (int rc = xfer_fields(b, fs.data(), offs.data(), (int)fs.size(), xbuf, false, true, off_status, off_resid)) { fprintf(stderr, "tiny_solve: %s\n", b->err); return rc; }
    }
    // download: status + residuals first (every group's scatter looks at the iteration counts), then the field groups, an
    // event behind each; the scatter of a group waits for its own event only
    const std::vector<size_t> ogb = split_groups(out, 0);
    const int ogroups = (int)ogb.size() - 1;
    for (int g = 0; g <= ogroups; ++g)
        if (!ctx.ev[g] && hipEventCreateWithFlags(&ctx.ev[g], hipEventDisableTiming) != hipSuccess) return TINY_ERR_HIP;
    if (!zero_copy && hipMemcpyAsync(ctx.h_pin + off_status, ctx.d_xfer + off_status, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return TINY_ERR_HIP;
    std::vector<size_t> gstart(ogroups + 1, 0);
    for (int g = 0; g < ogroups; ++g) {
        size_t o1 = gstart[g];
        for (size_t i = ogb[g]; i < ogb[g + 1]; ++i) o1 += (size_t)n * fsize(out[i]);
        gstart[g + 1] = o1;
        if ((!zero_copy && hipMemcpyAsync(ctx.h_pin + gstart[g], ctx.d_xfer + gstart[g], (o1 - gstart[g]) * sizeof(double), hipMemcpyDeviceToHost, b->stream) != hipSuccess) ||
            hipEventRecord(ctx.ev[g], b->stream) != hipSuccess) return TINY_ERR_HIP;
    }
    const double t2 = now();
    if (hipEventSynchronize(ctx.ev[0]) != hipSuccess) return TINY_ERR_HIP;
    const double t3 = now();
    const int4* st = reinterpret_cast<const int4*>(ctx.h_pin + off_status);
    const double* res = ctx.h_pin + off_resid;
    std::vector<std::string> lines(33);                       // "Solver converged ..." lines per thread, printed in solver order
    std::vector<int> unsolved(33, 0);
    for (int g = 0; g < ogroups; ++g) {
        if (g > 0 && hipEventSynchronize(ctx.ev[g]) != hipSuccess) return TINY_ERR_HIP;
        parallel_solvers(n, [&](int lo, int hi, int) {
            size_t o = gstart[g];
            for (size_t i = ogb[g]; i < ogb[g + 1]; ++i) {
                const FieldMap& fm = out[i];
                const size_t sz = fsize(fm);
                // max_iter = 0 (no iteration ran): the reference leaves x, u, q, r, p, d as they were
                const bool sweep_output = fm.f == TINY_F_X || fm.f == TINY_F_U || (fm.f >= TINY_F_Q && fm.f <= TINY_F_D);
                for (int k = lo; k < hi; ++k) {
                    if (sweep_output && st[k].x == 0) continue;
                    TinyMatrixPOD& m = solvers[k]->work->*(fm.m);
                    if ((size_t)(m.rows * m.cols) == sz) memcpy(m.data, ctx.h_pin + o + k * sz, sz * sizeof(double));
                }
                o += (size_t)n * sz;
            }
        });
    }
    if (hipStreamSynchronize(b->stream) != hipSuccess) return TINY_ERR_HIP;
    parallel_solvers(n, [&](int lo, int hi, int t) {
        for (int k = lo; k < hi; ++k) {
            TinySolver* s = solvers[k];
            TinyWorkspace* w = s->work;
            w->iter = st[k].x;                                   // admm.cpp:337,394
            w->status = st[k].z;                                 // admm.cpp:336,431
            if (st[k].w) {                                       // residuals are only written by a check (admm.cpp:312-317)
                w->primal_residual_state = res[4 * k + 0]; w->primal_residual_input = res[4 * k + 1];
                w->dual_residual_state = res[4 * k + 2]; w->dual_residual_input = res[4 * k + 3];
            }
            s->solution->iter = st[k].x;                         // admm.cpp:434-437 / :450-453
            s->solution->solved = st[k].y;
            mat_assign(&s->solution->x, w->vnew.data, w->vnew.rows, w->vnew.cols);
            mat_assign(&s->solution->u, w->znew.data, w->znew.rows, w->znew.cols);
            if (st[k].y) lines[t] += "Solver converged in " + std::to_string(w->iter) + " iterations\n";   // admm.cpp:439
            else unsolved[t] = 1;
        }
    });
    int all = 0;
    for (size_t t = 0; t < lines.size(); ++t) { std::cout << lines[t]; all |= unsolved[t]; }
    std::cout.flush();
    if (trace) fprintf(stderr, "tiny_solve trace (us): gather %.0f, enqueue %.0f, wait %.0f, scatter+print %.0f; in %zu B out %zu B\n", t1 - t0, t2 - t1, t3 - t2, now() - t3, in_doubles * 8, out_doubles * 8);
    return all;                                              // 0 converged / 1 max_iter (admm.cpp:441,454)
}

// One exported phase function of admm.hpp:12-17 on solver s: upload every workspace field the phases read, run the
// phase on the GPU, write back the fields that phase writes.  Returns < 0 on error, else the phase's boolean.
int phase_call(TinySolver* s, int phase) {
    if (!s || !s->work || !s->cache || !s->settings) return -TINY_ERR_NULL;
    const std::shared_ptr<Ctx> ctxp = context_of(s);
    std::lock_guard<std::mutex> lk(ctxp->mu);
    TinyBatch* b = nullptr;
    if (int rc = device_context(*ctxp, s, 1, &b)) { forget_context(s, ctxp); return -rc; }
    ctxp->family_valid = false;                        // the phase path re-uploads the family unconditionally
    if (int rc = sync_family(b, s)) { fprintf(stderr, "tinympc_amd phase: %s\n", b->err); return -rc; }
    TinyWorkspace* w = s->work;
    const TinySettings* st = s->settings;
    const int nx = w->nx, nu = w->nu, N = w->N;
    const size_t ns = (size_t)nx * N, ni = (size_t)nu * (N - 1);
    const bool s_soc = st->en_state_soc && w->numStateCones > 0, i_soc = st->en_input_soc && w->numInputCones > 0;
    struct Io { TinyField f; TinyMatrixPOD TinyWorkspace::*m; bool on; int writer; };   // writer: the phase that writes it
    const Io io[] = {
        {TINY_F_XREF, &TinyWorkspace::Xref, true, 0}, {TINY_F_UREF, &TinyWorkspace::Uref, true, 0},
        {TINY_F_X, &TinyWorkspace::x, true, PHASE_FORWARD}, {TINY_F_U, &TinyWorkspace::u, true, PHASE_FORWARD},
        {TINY_F_Q, &TinyWorkspace::q, true, PHASE_LINEAR_COST}, {TINY_F_R, &TinyWorkspace::r, true, PHASE_LINEAR_COST},
        {TINY_F_P, &TinyWorkspace::p, true, -1}, {TINY_F_D, &TinyWorkspace::d, true, PHASE_BACKWARD},
        {TINY_F_V, &TinyWorkspace::v, true, 0}, {TINY_F_Z, &TinyWorkspace::z, true, 0},
        {TINY_F_VNEW, &TinyWorkspace::vnew, true, PHASE_SLACK}, {TINY_F_ZNEW, &TinyWorkspace::znew, true, PHASE_SLACK},
        {TINY_F_G, &TinyWorkspace::g, true, PHASE_DUAL}, {TINY_F_Y, &TinyWorkspace::y, true, PHASE_DUAL},
        {TINY_F_VCNEW, &TinyWorkspace::vcnew, s_soc, PHASE_SLACK}, {TINY_F_ZCNEW, &TinyWorkspace::zcnew, i_soc, PHASE_SLACK},
        {TINY_F_GC, &TinyWorkspace::gc, s_soc, PHASE_DUAL}, {TINY_F_YC, &TinyWorkspace::yc, i_soc, PHASE_DUAL},
        {TINY_F_VLNEW, &TinyWorkspace::vlnew, st->en_state_linear != 0, PHASE_SLACK}, {TINY_F_ZLNEW, &TinyWorkspace::zlnew, st->en_input_linear != 0, PHASE_SLACK},
        {TINY_F_GL, &TinyWorkspace::gl, st->en_state_linear != 0, PHASE_DUAL}, {TINY_F_YL, &TinyWorkspace::yl, st->en_input_linear != 0, PHASE_DUAL},
        {TINY_F_VLNEW_TV, &TinyWorkspace::vlnew_tv, st->en_tv_state_linear != 0, PHASE_SLACK}, {TINY_F_ZLNEW_TV, &TinyWorkspace::zlnew_tv, st->en_tv_input_linear != 0, PHASE_SLACK},
        {TINY_F_GL_TV, &TinyWorkspace::gl_tv, st->en_tv_state_linear != 0, PHASE_DUAL}, {TINY_F_YL_TV, &TinyWorkspace::yl_tv, st->en_tv_input_linear != 0, PHASE_DUAL}};
    for (const Io& f : io) {
        if (!f.on) continue;
        const TinyMatrixPOD& m = w->*(f.m);
        const size_t sz = is_state_field(f.f) ? ns : ni;
        if ((size_t)(m.rows * m.cols) != sz || !m.data) { fprintf(stderr, "tinympc_amd phase: workspace field %d has the wrong size\n", (int)f.f); return -TINY_ERR_DIM; }
        if (int rc = tiny_batch_set(b, f.f, m.data, TINY_HOST)) { fprintf(stderr, "tinympc_amd phase: %s\n", b->err); return -rc; }
    }
    if (int rc = tiny_batch_phase(b, phase)) { fprintf(stderr, "tinympc_amd phase: %s\n", b->err); return -rc; }
    for (const Io& f : io) {
        // p is written by update_linear_cost (its last column) and by backward_pass_grad (the others)
        const bool written = f.on && (f.writer == phase || (f.writer == -1 && (phase == PHASE_LINEAR_COST || phase == PHASE_BACKWARD)));
        if (!written) continue;
        if (int rc = tiny_batch_get(b, f.f, (w->*(f.m)).data, TINY_HOST)) { fprintf(stderr, "tinympc_amd phase: %s\n", b->err); return -rc; }
    }
    if (phase != PHASE_TERMINATION) return 0;
    int4 st4;
    double res[4];
    if (hipMemcpyAsync(&st4, b->d_status, sizeof(int4), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return -TINY_ERR_HIP;
    if (hipMemcpyAsync(res, b->d_resid, sizeof(res), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return -TINY_ERR_HIP;
    if (hipStreamSynchronize(b->stream) != hipSuccess) return -TINY_ERR_HIP;
    w->primal_residual_state = res[0]; w->primal_residual_input = res[1];             // admm.cpp:314-317
    w->dual_residual_state = res[2]; w->dual_residual_input = res[3];
    return st4.y;
}

// a 1-thread GPU evaluation of the small projection utilities (no CPU arithmetic in this library)
int project_on_device(int which, double* v, const double* a, int n, float mu, double bb) {
    double *dv = nullptr, *da = nullptr;
    if (hipMalloc(&dv, (size_t)n * sizeof(double)) != hipSuccess) return TINY_ERR_HIP;
    int rc = TINY_OK;
    if (hipMemcpy(dv, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = TINY_ERR_HIP;
    if (!rc && a) {
        if (hipMalloc(&da, (size_t)n * sizeof(double)) != hipSuccess || hipMemcpy(da, a, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = TINY_ERR_HIP;
    }
    if (!rc) rc = launch_projection(which, dv, da, n, mu, bb);
    if (!rc && hipMemcpy(v, dv, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = TINY_ERR_HIP;
    hipFree(dv);
    if (da) hipFree(da);
    return rc;
}

}  // namespace

extern "C" {

int tiny_set_default_settings(TinySettings* settings) {           // tiny_api.cpp:413-441
    if (!settings) { std::cout << "Error in tiny_set_default_settings: settings is nullptr" << std::endl; return 1; }
    settings->abs_pri_tol = 1e-3; settings->abs_dua_tol = 1e-3;
    settings->max_iter = 1000; settings->check_termination = 1;
    settings->en_state_bound = 1; settings->en_input_bound = 1;
    settings->en_state_soc = 0; settings->en_input_soc = 0;
    settings->en_state_linear = 0; settings->en_input_linear = 0;
    settings->en_tv_state_linear = 0; settings->en_tv_input_linear = 0;
    settings->adaptive_rho = 0; settings->adaptive_rho_min = 1.0; settings->adaptive_rho_max = 100.0;
    settings->adaptive_rho_enable_clipping = 1;
    return 0;
}

int tiny_update_settings(TinySettings* settings, double abs_pri_tol, double abs_dua_tol, int max_iter,
                         int check_termination, int en_state_bound, int en_input_bound, int en_state_soc,
                         int en_input_soc, int en_state_linear, int en_input_linear, int en_tv_state_linear,
                         int en_tv_input_linear) {                 // tiny_api.cpp:388-411
    if (!settings) { std::cout << "Error in tiny_update_settings: settings is nullptr" << std::endl; return 1; }
    settings->abs_pri_tol = abs_pri_tol; settings->abs_dua_tol = abs_dua_tol;
    settings->max_iter = max_iter; settings->check_termination = check_termination;
    settings->en_state_bound = en_state_bound; settings->en_input_bound = en_input_bound;
    settings->en_state_soc = en_state_soc; settings->en_input_soc = en_input_soc;
    settings->en_state_linear = en_state_linear; settings->en_input_linear = en_input_linear;
    settings->en_tv_state_linear = en_tv_state_linear; settings->en_tv_input_linear = en_tv_input_linear;
    return 0;
}

int tiny_precompute_and_set_cache(TinyCache* cache, const TinyMatrixPOD* Adyn, const TinyMatrixPOD* Bdyn,
                                  const TinyMatrixPOD* fdyn, const TinyMatrixPOD* Q, const TinyMatrixPOD* R, int nx,
                                  int nu, double rho, int verbose) {   // tiny_api.cpp:307-381
    if (!cache) { std::cout << "Error in tiny_precompute_and_set_cache: cache is nullptr" << std::endl; return 1; }
    const Mat A = to_mat(Adyn), B = to_mat(Bdyn), f = to_mat(fdyn), Qm = to_mat(Q), Rm = to_mat(R);
    if (A.r != nx || A.c != nx || B.r != nx || B.c != nu || Qm.r != nx || Qm.c != nx || Rm.r != nu || Rm.c != nu ||
        f.r != nx || f.c != 1)
        return 1;
    if (verbose) {
        Mat Q1 = Qm, R1 = Rm;
        for (int i = 0; i < nx; ++i) Q1(i, i) += rho;
        for (int i = 0; i < nu; ++i) R1(i, i) += rho;
        std::cout << "A = " << fmt(A) << std::endl;
        std::cout << "B = " << fmt(B) << std::endl;
        std::cout << "Q = " << fmt(Q1) << std::endl;
        std::cout << "R = " << fmt(R1) << std::endl;
        std::cout << "rho = " << rho << std::endl;
    }
    Cache c;
    if (!precompute_cache(A, B, f, Qm, Rm, rho, &c)) return 1;
    if (verbose) {
        if (c.riccati_converged)      // printed only when the recursion converged (tiny_api.cpp:342-344)
            std::cout << "Kinf converged after " << c.riccati_iters << " iterations" << std::endl;
        std::cout << "Kinf = " << fmt(c.Kinf) << std::endl;
        std::cout << "Pinf = " << fmt(c.Pinf) << std::endl;
        std::cout << "Quu_inv = " << fmt(c.Quu_inv) << std::endl;
        std::cout << "AmBKt = " << fmt(c.AmBKt) << std::endl;
        std::cout << "APf = " << fmt(c.APf) << std::endl;
        std::cout << "BPf = " << fmt(c.BPf) << std::endl;
        std::cout << "\nPrecomputation finished!\n" << std::endl;
    }
    cache->rho = rho;
    mat_from(&cache->Kinf, c.Kinf); mat_from(&cache->Pinf, c.Pinf);
    mat_from(&cache->Quu_inv, c.Quu_inv); mat_from(&cache->AmBKt, c.AmBKt);
    mat_from(&cache->C1, c.Quu_inv); mat_from(&cache->C2, c.AmBKt);        // tiny_api.cpp:375-376
    vec_assign(&cache->APf, c.APf.a.data(), nx); vec_assign(&cache->BPf, c.BPf.a.data(), nu);
    return 0;
}

int tiny_setup(TinySolver** solverp, const TinyMatrixPOD* Adyn, const TinyMatrixPOD* Bdyn, const TinyMatrixPOD* fdyn,
               const TinyMatrixPOD* Q, const TinyMatrixPOD* R, double rho, int nx, int nu, int N, int verbose) {
    // tiny_api.cpp:21-147
    TinySolution* solution = (TinySolution*)calloc(1, sizeof(TinySolution));
    TinyCache* cache = (TinyCache*)calloc(1, sizeof(TinyCache));
    TinySettings* settings = (TinySettings*)calloc(1, sizeof(TinySettings));
    TinyWorkspace* work = (TinyWorkspace*)calloc(1, sizeof(TinyWorkspace));
    TinySolver* solver = (TinySolver*)calloc(1, sizeof(TinySolver));
    solver->solution = solution; solver->cache = cache; solver->settings = settings; solver->work = work;
    *solverp = solver;
    solution->iter = 0; solution->solved = 0;
    mat_alloc(&solution->x, nx, N); mat_alloc(&solution->u, nu, N - 1);
    tiny_set_default_settings(settings);
    work->nx = nx; work->nu = nu; work->N = N;
    int status = 0;
    status |= check_dimension("State transition matrix (A)", "rows", Adyn->rows, nx);
    status |= check_dimension("State transition matrix (A)", "columns", Adyn->cols, nx);
    status |= check_dimension("Input matrix (B)", "rows", Bdyn->rows, nx);
    status |= check_dimension("Input matrix (B)", "columns", Bdyn->cols, nu);
    status |= check_dimension("Affine vector (f)", "rows", fdyn->rows, nx);
    status |= check_dimension("Affine vector (f)", "columns", fdyn->cols, 1);
    status |= check_dimension("State stage cost (Q)", "rows", Q->rows, nx);
    status |= check_dimension("State stage cost (Q)", "columns", Q->cols, nx);
    status |= check_dimension("State input cost (R)", "rows", R->rows, nu);
    status |= check_dimension("State input cost (R)", "columns", R->cols, nu);
    if (status) return status;
    TinyMatrixPOD* st[] = {&work->x, &work->q, &work->p, &work->v, &work->vnew, &work->g, &work->vc, &work->vcnew,
                           &work->gc, &work->vl, &work->vlnew, &work->gl, &work->vl_tv, &work->vlnew_tv, &work->gl_tv,
                           &work->Xref};
    TinyMatrixPOD* in[] = {&work->u, &work->r, &work->d, &work->z, &work->znew, &work->y, &work->zc, &work->zcnew,
                           &work->yc, &work->zl, &work->zlnew, &work->yl, &work->zl_tv, &work->zlnew_tv, &work->yl_tv,
                           &work->Uref};
    for (TinyMatrixPOD* m : st) mat_alloc(m, nx, N);
    for (TinyMatrixPOD* m : in) mat_alloc(m, nu, N - 1);
    vec_alloc(&work->Q, nx); vec_alloc(&work->R, nu);
    for (int i = 0; i < nx; ++i) work->Q.data[i] = Q->data[(size_t)i * nx + i] + rho;     // :117
    for (int i = 0; i < nu; ++i) work->R.data[i] = R->data[(size_t)i * nu + i] + rho;     // :118
    mat_assign(&work->Adyn, Adyn->data, nx, nx);
    mat_assign(&work->Bdyn, Bdyn->data, nx, nu);
    vec_assign(&work->fdyn, fdyn->data, nx);
    vec_alloc(&work->Qu, nu);
    work->status = 0; work->iter = 0;
    // :136 -- the cache sees diag(work->Q), diag(work->R), i.e. rho a second time
    Mat Qd(nx, nx), Rd(nu, nu);
    for (int i = 0; i < nx; ++i) Qd(i, i) = work->Q.data[i];
    for (int i = 0; i < nu; ++i) Rd(i, i) = work->R.data[i];
    TinyMatrixPOD Qp{Qd.a.data(), nx, nx}, Rp{Rd.a.data(), nu, nu};
    TinyMatrixPOD fp{work->fdyn.data, nx, 1};
    status = tiny_precompute_and_set_cache(cache, &work->Adyn, &work->Bdyn, &fp, &Qp, &Rp, nx, nu, rho, verbose);
    return status;
}

int tiny_set_bound_constraints(TinySolver* solver, const TinyMatrixPOD* x_min, const TinyMatrixPOD* x_max,
                               const TinyMatrixPOD* u_min, const TinyMatrixPOD* u_max) {   // tiny_api.cpp:149-174
    if (!solver) { std::cout << "Error in tiny_set_bound_constraints: solver is nullptr" << std::endl; return 1; }
    TinyWorkspace* w = solver->work;
    int status = 0;
    status |= check_dimension("Lower state bounds (x_min)", "rows", x_min->rows, w->nx);
    status |= check_dimension("Lower state bounds (x_min)", "cols", x_min->cols, w->N);
    status |= check_dimension("Lower state bounds (x_max)", "rows", x_max->rows, w->nx);
    status |= check_dimension("Lower state bounds (x_max)", "cols", x_max->cols, w->N);
    status |= check_dimension("Lower input bounds (u_min)", "rows", u_min->rows, w->nu);
    status |= check_dimension("Lower input bounds (u_min)", "cols", u_min->cols, w->N - 1);
    status |= check_dimension("Lower input bounds (u_max)", "rows", u_max->rows, w->nu);
    status |= check_dimension("Lower input bounds (u_max)", "cols", u_max->cols, w->N - 1);
    mat_assign(&w->x_min, x_min->data, x_min->rows, x_min->cols);
    mat_assign(&w->x_max, x_max->data, x_max->rows, x_max->cols);
    mat_assign(&w->u_min, u_min->data, u_min->rows, u_min->cols);
    mat_assign(&w->u_max, u_max->data, u_max->rows, u_max->cols);
    return 0;                                                    // the reference ignores `status` here (:173)
}

int tiny_set_cone_constraints(TinySolver* solver, const TinyVectorXiPOD* Acx, const TinyVectorXiPOD* qcx,
                              const TinyVectorPOD* cx, const TinyVectorXiPOD* Acu, const TinyVectorXiPOD* qcu,
                              const TinyVectorPOD* cu) {           // tiny_api.cpp:176-208
    if (!solver) { std::cout << "Error in tiny_set_cone_constraints: solver is nullptr" << std::endl; return 1; }
    const int nsc = (int)Acx->rows, nic = (int)Acu->rows;
    int status = 0;
    status |= check_dimension("Cone state size (qcx)", "rows", qcx->rows, nsc);
    status |= check_dimension("Cone mu value for state (cx)", "rows", cx->rows, nsc);
    status |= check_dimension("Cone input size (qcu)", "rows", qcu->rows, nic);
    status |= check_dimension("Cone mu value for input (cu)", "rows", cu->rows, nic);
    if (status) return status;
    TinyWorkspace* w = solver->work;
    w->numStateCones = nsc; w->numInputCones = nic;
    ivec_assign(&w->Acx, Acx->data, nsc); ivec_assign(&w->qcx, qcx->data, nsc); vec_assign(&w->cx, cx->data, nsc);
    ivec_assign(&w->Acu, Acu->data, nic); ivec_assign(&w->qcu, qcu->data, nic); vec_assign(&w->cu, cu->data, nic);
    return 0;
}

int tiny_set_linear_constraints(TinySolver* solver, const TinyMatrixPOD* Alin_x, const TinyVectorPOD* blin_x,
                                const TinyMatrixPOD* Alin_u, const TinyVectorPOD* blin_u) {   // tiny_api.cpp:210-251
    if (!solver) { std::cout << "Error in tiny_set_linear_constraints: solver is nullptr" << std::endl; return 1; }
    TinyWorkspace* w = solver->work;
    const int nsl = (int)Alin_x->rows, nil = (int)Alin_u->rows;
    int status = 0;
    if (nsl > 0) {
        status |= check_dimension("State linear constraint matrix (Alin_x)", "columns", Alin_x->cols, w->nx);
        status |= check_dimension("State linear constraint vector (blin_x)", "rows", blin_x->rows, nsl);
    }
    if (nil > 0) {
        status |= check_dimension("Input linear constraint matrix (Alin_u)", "columns", Alin_u->cols, w->nu);
        status |= check_dimension("Input linear constraint vector (blin_u)", "rows", blin_u->rows, nil);
    }
    if (status) return status;
    w->numStateLinear = nsl; w->numInputLinear = nil;
    mat_assign(&w->Alin_x, Alin_x->data, Alin_x->rows, Alin_x->cols); vec_assign(&w->blin_x, blin_x->data, blin_x->rows);
    mat_assign(&w->Alin_u, Alin_u->data, Alin_u->rows, Alin_u->cols); vec_assign(&w->blin_u, blin_u->data, blin_u->rows);
    return 0;
}

int tiny_set_tv_linear_constraints(TinySolver* solver, const TinyMatrixPOD* tv_Alin_x, const TinyMatrixPOD* tv_blin_x,
                                   const TinyMatrixPOD* tv_Alin_u, const TinyMatrixPOD* tv_blin_u) {   // tiny_api.cpp:253-304
    if (!solver) { std::cout << "Error in tiny_set_linear_constraints: solver is nullptr" << std::endl; return 1; }
    TinyWorkspace* w = solver->work;
    const int nts = (int)(tv_Alin_x->rows / w->N), nti = (int)(tv_Alin_u->rows / (w->N - 1));
    int status = 0;
    if (nts > 0) {
        status |= check_dimension("State time-varying linear constraint matrix (tv_Alin_x)", "rows", tv_Alin_x->rows, (long)nts * w->N);
        status |= check_dimension("State time-varying linear constraint matrix (tv_Alin_x)", "columns", tv_Alin_x->cols, w->nx);
        status |= check_dimension("State time-varying linear constraint vector (tv_blin_x)", "rows", tv_blin_x->rows, nts);
        status |= check_dimension("State time-varying linear constraint vector (tv_blin_x)", "columns", tv_blin_x->cols, w->N);
    }
    if (nti > 0) {
        status |= check_dimension("Input time-varying linear constraint matrix (tv_Alin_u)", "rows", tv_Alin_u->rows, (long)nti * (w->N - 1));
        status |= check_dimension("Input time-varying linear constraint matrix (tv_Alin_u)", "columns", tv_Alin_u->cols, w->nu);
        status |= check_dimension("Input time-varying linear constraint vector (tv_blin_u)", "rows", tv_blin_u->rows, nti);
        status |= check_dimension("Input time-varying linear constraint vector (tv_blin_u)", "columns", tv_blin_u->cols, w->N - 1);
    }
    if (status) return status;
    w->numtvStateLinear = nts; w->numtvInputLinear = nti;
    mat_assign(&w->tv_Alin_x, tv_Alin_x->data, tv_Alin_x->rows, tv_Alin_x->cols);
    mat_assign(&w->tv_blin_x, tv_blin_x->data, tv_blin_x->rows, tv_blin_x->cols);
    mat_assign(&w->tv_Alin_u, tv_Alin_u->data, tv_Alin_u->rows, tv_Alin_u->cols);
    mat_assign(&w->tv_blin_u, tv_blin_u->data, tv_blin_u->rows, tv_blin_u->cols);
    return 0;
}

int tiny_set_x0(TinySolver* solver, const TinyVectorPOD* x0) {     // tiny_api.cpp:443-453
    if (!solver) { std::cout << "Error in tiny_set_x0: solver is nullptr" << std::endl; return 1; }
    if (x0->rows != solver->work->nx) { perror("Error in tiny_set_x0: x0 is not the correct length"); return 0; }
    memcpy(solver->work->x.data, x0->data, (size_t)x0->rows * sizeof(double));   // x.col(0) = x0
    return 0;
}

int tiny_set_x_ref(TinySolver* solver, const TinyMatrixPOD* x_ref) {   // tiny_api.cpp:455-465
    if (!solver) { std::cout << "Error in tiny_set_x_ref: solver is nullptr" << std::endl; return 1; }
    check_dimension("State reference trajectory (x_ref)", "rows", x_ref->rows, solver->work->nx);
    check_dimension("State reference trajectory (x_ref)", "columns", x_ref->cols, solver->work->N);
    mat_assign(&solver->work->Xref, x_ref->data, x_ref->rows, x_ref->cols);
    return 0;
}

int tiny_set_u_ref(TinySolver* solver, const TinyMatrixPOD* u_ref) {   // tiny_api.cpp:467-477
    if (!solver) { std::cout << "Error in tiny_set_u_ref: solver is nullptr" << std::endl; return 1; }
    check_dimension("Control/input reference trajectory (u_ref)", "rows", u_ref->rows, solver->work->nu);
    check_dimension("Control/input reference trajectory (u_ref)", "columns", u_ref->cols, solver->work->N - 1);
    mat_assign(&solver->work->Uref, u_ref->data, u_ref->rows, u_ref->cols);
    return 0;
}

int solve(TinySolver* solver) { return solve_group(&solver, 1); }          // admm.cpp:331
int tiny_solve(TinySolver* solver) { return solve(solver); }               // tiny_api.cpp:384-386
// tiny_api.hpp:54 / tiny_api.cpp:479-540: the quadrotor's hard-coded d(.)/d(rho) tables.  The reference assigns fixed-size
// 4 x 12 / 12 x 12 / 4 x 4 / 12 x 12 maps whatever the solver's nx, nu are; so does this (the values are the ones the real
// function leaves behind -- read through column-major maps over row-major literals --, kept as data: tinympc_amd/data/).
void tiny_initialize_sensitivity_matrices(TinySolver* solver) {
    if (!solver || !solver->cache) return;
    TinyCache* c = solver->cache;
    mat_assign(&c->dKinf_drho, k_dKinf_drho, 4, 12);
    mat_assign(&c->dPinf_drho, k_dPinf_drho, 12, 12);
    mat_assign(&c->dC1_drho, k_dC1_drho, 4, 4);
    mat_assign(&c->dC2_drho, k_dC2_drho, 12, 12);
}

int tiny_solve_batch(TinySolver** solvers, int n) { return solve_group(solvers, n); }

// ---- admm.hpp:12-17: the phases of one iteration, each on the GPU ------------------------------
void update_linear_cost(TinySolver* solver) { phase_call(solver, PHASE_LINEAR_COST); }      // admm.cpp:262
void backward_pass_grad(TinySolver* solver) { phase_call(solver, PHASE_BACKWARD); }         // admm.cpp:13
void forward_pass(TinySolver* solver) { phase_call(solver, PHASE_FORWARD); }                // admm.cpp:25
void update_slack(TinySolver* solver) { phase_call(solver, PHASE_SLACK); }                  // admm.cpp:81
void update_dual(TinySolver* solver) { phase_call(solver, PHASE_DUAL); }                    // admm.cpp:219
bool termination_condition(TinySolver* solver) {                                             // admm.cpp:310
    if (!solver || !solver->work || !solver->settings) return false;
    const int ct = solver->settings->check_termination;
    if (ct == 0 || solver->work->iter % ct != 0) return false;      // :312 (the reference divides by zero for ct == 0)
    return phase_call(solver, PHASE_TERMINATION) == 1;
}
TinyVectorPOD* project_soc(TinyVectorPOD* result, const TinyVectorPOD* s, float mu) {        // admm.cpp:39
    result->data = nullptr; result->rows = 0;
    vec_assign(result, s->data, s->rows);
    if (s->rows > 0 && project_on_device(0, result->data, nullptr, (int)s->rows, mu, 0.0) != TINY_OK)
        fprintf(stderr, "project_soc: GPU evaluation failed (libtinympc_amd has no CPU path)\n");
    return result;
}
TinyVectorPOD* project_hyperplane(TinyVectorPOD* result, const TinyVectorPOD* z, const TinyVectorPOD* a, double b) {   // admm.cpp:70
    result->data = nullptr; result->rows = 0;
    vec_assign(result, z->data, z->rows);
    if (z->rows > 0 && project_on_device(1, result->data, a->data, (int)z->rows, 0.0f, b) != TINY_OK)
        fprintf(stderr, "project_hyperplane: GPU evaluation failed (libtinympc_amd has no CPU path)\n");
    return result;
}

int tiny_destroy(TinySolver* solver) {
    if (!solver) return TINY_ERR_NULL;
    {
        std::shared_ptr<Ctx> c;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            auto it = g_ctx.find(solver);
            if (it != g_ctx.end()) { c = it->second; g_ctx.erase(it); }
        }
        if (c) { std::lock_guard<std::mutex> lk(c->mu); release(*c); }
    }
    // every matrix member is {data*, ...}: walk the structs as arrays of words would be fragile; free by name
    TinyWorkspace* w = solver->work;
    TinyMatrixPOD* ms[] = {&w->x, &w->u, &w->q, &w->r, &w->p, &w->d, &w->v, &w->vnew, &w->z, &w->znew, &w->g, &w->y,
                           &w->x_min, &w->x_max, &w->u_min, &w->u_max, &w->vc, &w->vcnew, &w->zc, &w->zcnew, &w->gc,
                           &w->yc, &w->Alin_x, &w->Alin_u, &w->vl, &w->vlnew, &w->zl, &w->zlnew, &w->gl, &w->yl,
                           &w->tv_Alin_x, &w->tv_blin_x, &w->tv_Alin_u, &w->tv_blin_u, &w->vl_tv, &w->vlnew_tv,
                           &w->zl_tv, &w->zlnew_tv, &w->gl_tv, &w->yl_tv, &w->Adyn, &w->Bdyn, &w->Xref, &w->Uref,
                           &solver->solution->x, &solver->solution->u, &solver->cache->Kinf, &solver->cache->Pinf,
                           &solver->cache->Quu_inv, &solver->cache->AmBKt, &solver->cache->C1, &solver->cache->C2,
                           &solver->cache->dKinf_drho, &solver->cache->dPinf_drho, &solver->cache->dC1_drho,
                           &solver->cache->dC2_drho};
    for (TinyMatrixPOD* m : ms) free(m->data);
    TinyVectorPOD* vs[] = {&w->cx, &w->cu, &w->blin_x, &w->blin_u, &w->Q, &w->R, &w->fdyn, &w->Qu, &solver->cache->APf,
                           &solver->cache->BPf};
    for (TinyVectorPOD* v : vs) free(v->data);
    free(w->Acx.data); free(w->Acu.data); free(w->qcx.data); free(w->qcu.data);
    free(solver->solution); free(solver->cache); free(solver->settings); free(solver->work); free(solver);
    return 0;
}

}  // extern "C"

// layout contract with the reference (measured with offsetof on its types.hpp)
static_assert(sizeof(TinyMatrixPOD) == 24 && sizeof(TinyVectorPOD) == 16 && sizeof(TinyVectorXiPOD) == 16, "Eigen DenseStorage");
static_assert(sizeof(TinySolution) == 56 && offsetof(TinySolution, x) == 8 && offsetof(TinySolution, u) == 32, "TinySolution");
static_assert(sizeof(TinyCache) == 280 && offsetof(TinyCache, Kinf) == 8 && offsetof(TinyCache, APf) == 104 &&
              offsetof(TinyCache, BPf) == 120 && offsetof(TinyCache, C1) == 136 && offsetof(TinyCache, dC2_drho) == 256, "TinyCache");
static_assert(sizeof(TinySettings) == 88 && offsetof(TinySettings, max_iter) == 16 && offsetof(TinySettings, adaptive_rho) == 56 &&
              offsetof(TinySettings, adaptive_rho_min) == 64 && offsetof(TinySettings, adaptive_rho_enable_clipping) == 80, "TinySettings");
static_assert(sizeof(TinyWorkspace) == 1328 && offsetof(TinyWorkspace, x) == 16 && offsetof(TinyWorkspace, x_min) == 304 &&
              offsetof(TinyWorkspace, numStateCones) == 400 && offsetof(TinyWorkspace, cx) == 408 && offsetof(TinyWorkspace, vc) == 504 &&
              offsetof(TinyWorkspace, numStateLinear) == 648 && offsetof(TinyWorkspace, Alin_x) == 656 && offsetof(TinyWorkspace, vl) == 736 &&
              offsetof(TinyWorkspace, numtvStateLinear) == 880 && offsetof(TinyWorkspace, vl_tv) == 984 && offsetof(TinyWorkspace, Q) == 1128 &&
              offsetof(TinyWorkspace, Adyn) == 1160 && offsetof(TinyWorkspace, Xref) == 1224 && offsetof(TinyWorkspace, Qu) == 1272 &&
              offsetof(TinyWorkspace, primal_residual_state) == 1288 && offsetof(TinyWorkspace, status) == 1320 &&
              offsetof(TinyWorkspace, iter) == 1324, "TinyWorkspace");
static_assert(sizeof(TinySolver) == 32 && offsetof(TinySolver, work) == 24, "TinySolver");
