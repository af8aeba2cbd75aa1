/* examples/batched_double_integrator.c -- the batched, device-resident C ABI from plain C99.
 *
 * 4 096 planar double integrators (nx = 4: position / velocity in x and y, nu = 2: accelerations, N = 10, dt = 0.1 s)
 * are driven to the origin in closed loop: 60 MPC steps fused into 3 kernel launches, plant stepped on the GPU.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/batched_double_integrator.c -Ltinympc_amd -ltinympc_amd \
 *       -Wl,-rpath,$PWD/tinympc_amd -o batched_double_integrator && ./batched_double_integrator
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "tinympc_amd.h"

#define NX 4
#define NU 2
#define NH 10
#define BATCH 4096

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, tiny_batch_last_error(h)); return 1; } \
    } while (0)

int main(void) {
    const double dt = 0.1;
    /* column-major A (nx x nx), B (nx x nu): states (px, py, vx, vy) */
    double A[NX * NX] = {0}, B[NX * NU] = {0}, Q[NX] = {10, 10, 1, 1}, R[NU] = {0.5, 0.5};
    for (int i = 0; i < NX; ++i) A[i + NX * i] = 1.0;
    A[0 + NX * 2] = dt; A[1 + NX * 3] = dt;
    B[0 + NX * 0] = 0.5 * dt * dt; B[2 + NX * 0] = dt;
    B[1 + NX * 1] = 0.5 * dt * dt; B[3 + NX * 1] = dt;

    TinyBatch* h = NULL;
    int rc = tiny_batch_setup(&h, A, B, NULL, Q, R, 1.0, NX, NU, NH, BATCH, 0, 0);
    if (rc) { fprintf(stderr, "tiny_batch_setup failed (%d): no MI355X?\n", rc); return 1; }

    /* box constraints, replicated over the horizon: |v| <= 2 m/s, |a| <= 1 m/s^2 */
    double xmin[NX * NH], xmax[NX * NH], umin[NU * (NH - 1)], umax[NU * (NH - 1)];
    for (int k = 0; k < NH; ++k)
        for (int i = 0; i < NX; ++i) { xmax[i + NX * k] = i < 2 ? 1e17 : 2.0; xmin[i + NX * k] = -xmax[i + NX * k]; }
    for (int e = 0; e < NU * (NH - 1); ++e) { umax[e] = 1.0; umin[e] = -1.0; }
    CHECK(tiny_batch_set_bound_constraints(h, xmin, xmax, umin, umax));
    CHECK(tiny_batch_update_settings(h, 1e-3, 1e-3, 100, 1, 1, 1, 0, 0, 0, 0, 0, 0));

    /* every instance starts somewhere else */
    double* x0 = (double*)malloc(sizeof(double) * BATCH * NX);
    srand(7);
    for (int b = 0; b < BATCH; ++b) {
        x0[b * NX + 0] = 4.0 * rand() / RAND_MAX - 2.0;
        x0[b * NX + 1] = 4.0 * rand() / RAND_MAX - 2.0;
        x0[b * NX + 2] = x0[b * NX + 3] = 0.0;
    }
    CHECK(tiny_batch_set(h, TINY_F_X0, x0, TINY_HOST));
    CHECK(tiny_batch_set_option(h, "steps_per_launch", 20));     /* 20 closed-loop MPC steps per kernel launch */
    for (int launch = 0; launch < 3; ++launch) CHECK(tiny_batch_solve_async(h));
    CHECK(tiny_batch_synchronize(h));

    double stats[10];
    CHECK(tiny_batch_reduce_stats(h, stats, NULL));
    CHECK(tiny_batch_get(h, TINY_F_X0, x0, TINY_HOST));            /* the plant state after 60 steps */
    double worst = 0.0;
    for (int b = 0; b < BATCH; ++b) {
        const double d = hypot(x0[b * NX + 0], x0[b * NX + 1]);
        if (d > worst) worst = d;
    }
    printf("%d instances x 60 MPC steps: %.0f ADMM iterations, %.0f of %d solves converged, kernel path %d\n", BATCH, stats[7],
           stats[8], BATCH * 60, tiny_batch_kernel_path(h));
    printf("largest distance from the origin after 6 s: %.4f m\n", worst);
    free(x0);
    tiny_batch_destroy(h);
    return worst < 0.2 ? 0 : 2;
}
