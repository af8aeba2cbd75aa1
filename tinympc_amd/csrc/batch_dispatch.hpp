// batch_dispatch.hpp -- what batch_api.hip (the C ABI), batch_dispatch.hip (kernel selection, launch forms), batch_tables.hip (lane
// tables) and batch_helpers.hip (helper kernels, cost model, on-demand buffers) share.
#pragma once
#include "batch_impl.hpp"

#define HIP_TRY(b, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (void)hipGetLastError();   /* reported here: do not let it resurface in a later, unrelated call */ \
            return fail(b, TINY_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_));               \
        }                                                                                         \
    } while (0)

namespace tinympc_amd {
// ---- the compiled-in kernels (one translation unit per shape, _gen/registry.inc)
extern const KernelEntry* const* const g_kernel_list;
extern const int g_nkernels;
const TileEntry* find_tile(int nx, int nu, int N);
const KernelEntry* find_kernel(int nx, int nu, int N);
// ---- which kernel serves the batch as it is configured now
bool has_regs(const TinyBatch* b);
bool soc_active(const TinyBatch* b);
bool linear_active(const TinyBatch* b);
int lin_variant(const TinyBatch* b);
bool use_tile(const TinyBatch* b);
bool use_general(const TinyBatch* b);
bool box_is_uniform(const TinyBatch* b);     // the box is the same at every knot (from the host copies of the bounds)
// ---- buffers and state the API calls create on demand
int ensure_kpi(TinyBatch* b, double** p);
int ensure_repack_buffers(TinyBatch* b);
int ensure_regroup_buffers(TinyBatch* b, bool second_stream);
int ensure_adaptive(TinyBatch* b, bool need_tables = true);
int adaptive_fresh_state(TinyBatch* b);
// ---- lane tables (batch_tables.hip)
int lin_kmax(const TinyBatch* b);       // half-spaces per knot and family the LIN variants are built for (4, 8, 16, 32; 0: coverage kernel)
void build_tables(TinyBatch* b);
void build_tile_tables(TinyBatch* b);
void build_general_tables(TinyBatch* b);
int upload_tables(TinyBatch* b);
// ---- helper kernels, cost model (batch_helpers.hip)
constexpr int REGROUP_AUTO_MIN_STEPS = 16, REGROUP_AUTO_MIN_BATCH = 4096;
double wave_iteration_us(int nx, int nu, int N, int wps);
int enqueue_iteration_histogram(TinyBatch* b);
int enqueue_regroup_sort(TinyBatch* b, hipStream_t st, int half, int first, int count);
int enqueue_repack_sort(TinyBatch* b, const int* list, const int* count);
int enqueue_lockstep_estimate(TinyBatch* b);
// ---- launches besides launch_solve (batch_impl.hpp)
int launch_general(TinyBatch* b, int phase = 0);
int launch_riccati(TinyBatch* b, const RiccatiArgs& r, size_t lds_bytes, int grid);
// ---- the clock-checked dispatch: what probes left behind, the cost models as host arithmetic
void learn_from_probe(TinyBatch* b, int max_iter, bool auto_split);
void read_lockstep_estimate(TinyBatch* b);
int choose_split(const TinyBatch* b, const unsigned* hist, double* ratio);
int choose_split_for(int nx, int nu, int N, bool soc, int M, int ct, int gr, int num_cus, const unsigned* hist, double* ratio, int* growth_out = nullptr);
std::vector<int> regroup_stretches(int steps, int K, int lead);
int regroup_auto_k(int steps);
bool regroup_two_streams_apply(int steps, int lead, int K);
}  // namespace tinympc_amd
