/* examples/hetero_tracking.c -- round-5 surface of the batched C ABI from plain C99: per-instance problem data on a WIDE shape
 * (nx = 18, nu = 6: three planar double integrators chained per instance, every instance with its own time step and its own input
 * cost, N = 10 -- the tile kernel's per-instance form), a shared reference trajectory whose window moves one knot per MPC step
 * (the caller pattern of the reference's examples/quadrotor_tracking.cpp:77-106) with the duals reset before every solve, 40 MPC
 * steps fused into 4 launches, and the settled launch plan of a batch exported and imported into a second handle.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/hetero_tracking.c -Ltinympc_amd -ltinympc_amd -Wl,-rpath,$PWD/tinympc_amd -lm -o hetero_tracking
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tinympc_amd.h"

#define NX 18
#define NU 6
#define NH 10
#define BATCH 2048
#define STEPS 40
#define POINTS (NH + STEPS + 1)

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, h ? tiny_batch_last_error(h) : "-"); return 1; } \
    } while (0)

int main(void) {
    /* per-instance families: a chain of three planar double integrators (state block c: px, py, vx, vy + two coupling states), dt and R differ */
    double* A = (double*)calloc((size_t)BATCH * NX * NX, sizeof(double));
    double* B = (double*)calloc((size_t)BATCH * NX * NU, sizeof(double));
    double* Q = (double*)malloc(sizeof(double) * BATCH * NX);
    double* R = (double*)malloc(sizeof(double) * BATCH * NU);
    double* rho = (double*)malloc(sizeof(double) * BATCH);
    srand(11);
    for (int b = 0; b < BATCH; ++b) {
        const double dt = 0.08 + 0.04 * rand() / RAND_MAX;
        double* Ab = A + (size_t)b * NX * NX;
        double* Bb = B + (size_t)b * NX * NU;
        for (int i = 0; i < NX; ++i) Ab[i + NX * i] = i % 6 < 4 ? 1.0 : 0.9;          /* column-major */
        for (int c = 0; c < 3; ++c) {
            const int o = 6 * c;
            Ab[(o + 0) + NX * (o + 2)] = dt; Ab[(o + 1) + NX * (o + 3)] = dt;          /* p += dt v */
            Ab[(o + 4) + NX * (o + 0)] = 0.05; Ab[(o + 5) + NX * (o + 1)] = 0.05;      /* two lagged copies of the position */
            Bb[(o + 0) + NX * (2 * c + 0)] = 0.5 * dt * dt; Bb[(o + 2) + NX * (2 * c + 0)] = dt;
            Bb[(o + 1) + NX * (2 * c + 1)] = 0.5 * dt * dt; Bb[(o + 3) + NX * (2 * c + 1)] = dt;
        }
        for (int i = 0; i < NX; ++i) Q[(size_t)b * NX + i] = i % 6 < 2 ? 10.0 : (i % 6 < 4 ? 1.0 : 0.1);
        for (int a = 0; a < NU; ++a) R[(size_t)b * NU + a] = 0.2 + 0.6 * rand() / RAND_MAX;
        rho[b] = 1.0;
    }
    TinyBatch* h = NULL;
    int rc = tiny_batch_setup_hetero(&h, A, B, NULL, Q, R, rho, NX, NU, NH, BATCH, 0, 0);
    if (rc) { fprintf(stderr, "tiny_batch_setup_hetero failed (%d): no MI355X, or no hipRTC for the wide shape's per-instance form?\n", rc); return 1; }

    double xmin[NX * NH], xmax[NX * NH], umin[NU * (NH - 1)], umax[NU * (NH - 1)];
    for (int e = 0; e < NX * NH; ++e) { xmax[e] = 1e17; xmin[e] = -1e17; }
    for (int e = 0; e < NU * (NH - 1); ++e) { umax[e] = 2.0; umin[e] = -2.0; }
    CHECK(tiny_batch_set_bound_constraints(h, xmin, xmax, umin, umax));
    CHECK(tiny_batch_update_settings(h, 1e-3, 1e-3, 60, 1, 1, 1, 0, 0, 0, 0, 0, 0));

    /* the shared reference: the three bodies move along x at 0.5 m/s, one metre apart in y; window k .. k + N - 1 at MPC step k */
    double traj[POINTS * NX];
    memset(traj, 0, sizeof(traj));
    for (int k = 0; k < POINTS; ++k)
        for (int c = 0; c < 3; ++c) { traj[k * NX + 6 * c + 0] = 0.05 * k; traj[k * NX + 6 * c + 1] = (double)c; traj[k * NX + 6 * c + 2] = 0.5; }
    CHECK(tiny_batch_set_reference_trajectory(h, traj, POINTS, NULL, TINY_HOST));
    CHECK(tiny_batch_set_option(h, "reset_duals", 1));
    CHECK(tiny_batch_set_option(h, "advance_x0", 1));
    CHECK(tiny_batch_set_option(h, "steps_per_launch", STEPS / 4));

    double* x0 = (double*)calloc((size_t)BATCH * NX, sizeof(double));
    for (int b = 0; b < BATCH; ++b)
        for (int c = 0; c < 3; ++c) { x0[(size_t)b * NX + 6 * c + 0] = 0.6 * rand() / RAND_MAX - 0.3; x0[(size_t)b * NX + 6 * c + 1] = c + 0.6 * rand() / RAND_MAX - 0.3; }
    CHECK(tiny_batch_set(h, TINY_F_X0, x0, TINY_HOST));
    const int path = tiny_batch_kernel_path(h);
    for (int launch = 0; launch < 4; ++launch) CHECK(tiny_batch_solve_async(h));
    CHECK(tiny_batch_synchronize(h));
    double stats[10];
    CHECK(tiny_batch_reduce_stats(h, stats, NULL));
    CHECK(tiny_batch_get(h, TINY_F_X0, x0, TINY_HOST));
    double worst = 0.0;                                  /* distance of every body from the reference point the window has reached */
    for (int b = 0; b < BATCH; ++b)
        for (int c = 0; c < 3; ++c) {
            const double d = hypot(x0[(size_t)b * NX + 6 * c + 0] - 0.05 * STEPS, x0[(size_t)b * NX + 6 * c + 1] - c);
            if (d > worst) worst = d;
        }
    printf("%d families (nx = %d, nu = %d, N = %d) x %d MPC steps on a moving reference window: %.0f ADMM iterations, %.0f of %d solves converged, kernel path %d\n",
           BATCH, NX, NU, NH, STEPS, stats[7], stats[8], BATCH * STEPS, path);
    printf("largest tracking error after %d steps: %.4f m\n", STEPS, worst);

    /* the launch plan as plain data: what this handle's dispatch has settled on, into a fresh handle of the same shape */
    TinyBatchPlan plan;
    CHECK(tiny_batch_get_plan(h, &plan));
    tiny_batch_destroy(h);
    h = NULL;
    rc = tiny_batch_setup_hetero(&h, A, B, NULL, Q, R, rho, NX, NU, NH, BATCH, 0, 0);
    if (rc) { fprintf(stderr, "second setup failed (%d)\n", rc); return 1; }
    CHECK(tiny_batch_set_plan(h, &plan));
    printf("plan: %d bytes, shape (%d,%d,%d), open questions %d -- imported into a second handle\n", plan.bytes, plan.nx, plan.nu, plan.N, plan.open_questions);
    tiny_batch_destroy(h);
    free(A); free(B); free(Q); free(R); free(rho); free(x0);
    return (worst < 0.25 && (path == 1 || path == 4)) ? 0 : 2;     /* 1: tile kernel, compiled-in shape; 4: tile kernel, shape instantiated at run time */
}
