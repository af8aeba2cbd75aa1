/* examples/multi_gpu_rccl.c -- the caller-owned-communicator form of the statistics exchange.
 *
 * Hosts that already run one rank per GPU (MPI, torchrun, one thread per device) keep their own RCCL communicator;
 * tiny_batch_allreduce_stats(batch, comm, n_ranks, rank, total_batch, out) performs the path's one exchange on it: a
 * 64-byte message per rank in ONE ncclAllGather, reduced on the host.  Here the ranks are pthreads of one process and
 * the communicators come from ncclCommInitAll; with MPI the only change is ncclCommInitRank + a broadcast ncclUniqueId.
 *
 *   hipcc -x c -std=c99 ... or:
 *   gcc -std=gnu99 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/multi_gpu_rccl.c -Ltinympc_amd -ltinympc_amd \
 *       -L/opt/rocm/lib -lrccl -lamdhip64 -lpthread -lm -Wl,-rpath,$PWD/tinympc_amd -Wl,-rpath,/opt/rocm/lib -o multi_gpu_rccl
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

#include <rccl/rccl.h>

#include "tinympc_amd.h"

#define NX 4
#define NU 2
#define NH 10
#define MAX_GPUS 16

typedef struct {
    int rank, n_ranks, per_gpu, rc;
    ncclComm_t comm;
    double stats[10];
} Rank;

static void* rank_main(void* arg) {
    Rank* r = (Rank*)arg;
    const double dt = 0.1;
    double A[NX * NX] = {0}, B[NX * NU] = {0}, Q[NX] = {10, 10, 1, 1}, R[NU] = {0.5, 0.5};
    for (int i = 0; i < NX; ++i) A[i + NX * i] = 1.0;
    A[0 + NX * 2] = dt; A[1 + NX * 3] = dt;
    B[0 + NX * 0] = 0.5 * dt * dt; B[2 + NX * 0] = dt;
    B[1 + NX * 1] = 0.5 * dt * dt; B[3 + NX * 1] = dt;
    TinyBatch* h = NULL;
    r->rc = tiny_batch_setup(&h, A, B, NULL, Q, R, 1.0, NX, NU, NH, r->per_gpu, r->rank, 0);     /* this rank's shard on its GPU */
    if (r->rc) return NULL;
    double xmin[NX * NH], xmax[NX * NH], umin[NU * (NH - 1)], umax[NU * (NH - 1)];
    for (int k = 0; k < NH; ++k)
        for (int i = 0; i < NX; ++i) { xmax[i + NX * k] = i < 2 ? 1e17 : 2.0; xmin[i + NX * k] = -xmax[i + NX * k]; }
    for (int e = 0; e < NU * (NH - 1); ++e) { umax[e] = 1.0; umin[e] = -1.0; }
    tiny_batch_set_bound_constraints(h, xmin, xmax, umin, umax);
    tiny_batch_update_settings(h, 1e-3, 1e-3, 100, 1, 1, 1, 0, 0, 0, 0, 0, 0);
    double* x0 = (double*)malloc(sizeof(double) * (size_t)r->per_gpu * NX);
    unsigned seed = 7u + (unsigned)r->rank;
    for (int b = 0; b < r->per_gpu; ++b) {
        x0[b * NX + 0] = 4.0 * rand_r(&seed) / RAND_MAX - 2.0;
        x0[b * NX + 1] = 4.0 * rand_r(&seed) / RAND_MAX - 2.0;
        x0[b * NX + 2] = x0[b * NX + 3] = 0.0;
    }
    tiny_batch_set(h, TINY_F_X0, x0, TINY_HOST);
    tiny_batch_set_option(h, "steps_per_launch", 20);
    for (int launch = 0; launch < 3; ++launch) tiny_batch_solve_async(h);
    /* the one exchange of the path, on the caller's communicator */
    r->rc = tiny_batch_allreduce_stats(h, (void*)r->comm, r->n_ranks, r->rank, (long)r->per_gpu * r->n_ranks, r->stats);
    if (r->rc) fprintf(stderr, "rank %d: %s\n", r->rank, tiny_batch_last_error(h));
    free(x0);
    tiny_batch_destroy(h);
    return NULL;
}

int main(int argc, char** argv) {
    int gpus = tiny_batch_device_count();
    if (gpus <= 0) { fprintf(stderr, "no MI355X: libtinympc_amd has no CPU path\n"); return 1; }
    if (gpus > MAX_GPUS) gpus = MAX_GPUS;
    const int per_gpu = argc > 1 ? atoi(argv[1]) : 16384;
    ncclComm_t comms[MAX_GPUS];
    int devs[MAX_GPUS];
    for (int i = 0; i < gpus; ++i) devs[i] = i;
    if (ncclCommInitAll(comms, gpus, devs) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    Rank ranks[MAX_GPUS];
    pthread_t th[MAX_GPUS];
    for (int i = 0; i < gpus; ++i) {
        ranks[i].rank = i; ranks[i].n_ranks = gpus; ranks[i].per_gpu = per_gpu; ranks[i].comm = comms[i]; ranks[i].rc = 0;
        pthread_create(&th[i], NULL, rank_main, &ranks[i]);
    }
    int bad = 0;
    for (int i = 0; i < gpus; ++i) { pthread_join(th[i], NULL); bad |= ranks[i].rc; }
    for (int i = 0; i < gpus; ++i) ncclCommDestroy(comms[i]);
    if (bad) return 1;
    for (int i = 1; i < gpus; ++i)
        for (int k = 0; k < 10; ++k)
            if (ranks[i].stats[k] != ranks[0].stats[k]) { fprintf(stderr, "rank %d disagrees on entry %d\n", i, k); return 2; }
    const double* s = ranks[0].stats;
    printf("%d rank(s): %.0f instances x 60 MPC steps, %.0f ADMM iterations, %.0f of %.0f solves converged, max primal residual %.3e\n",
           gpus, s[2], s[7], s[8], 60.0 * s[2], fmax(s[3], s[4]));
    return (s[2] == (double)per_gpu * gpus && s[7] > 0.0 && s[8] > 0.9 * 60.0 * s[2]) ? 0 : 3;
}
