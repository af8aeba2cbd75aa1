/* dropin_threads.c -- the reference's operator API (tiny_setup / tiny_set_x0 / tiny_solve: tiny_api.hpp:10-47) called from several
 * host threads at once, every thread on its OWN TinySolver.  The reference is re-entrant there -- its only process-wide state is
 * the print format (tiny_api.cpp:11) -- and so is this library since round 6: a solver's device context carries its own lock, the
 * process-wide one only guards the map of contexts.
 *
 *   dropin_threads [threads] [steps]      (default 8 threads x 200 closed-loop MPC steps, two problem families)
 *
 * Every solver first runs its episode ALONE (serial pass), then all of them run the same episode concurrently; the program prints
 * one line per solver -- the iteration total and the final state must be identical bit for bit -- and the throughput ratio. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tinympc_amd.h"

#define NMAX 8
typedef struct {
    int nx, nu, N;
    double A[NMAX * NMAX], B[NMAX * NMAX], Q[NMAX], R[NMAX];   /* column-major A (nx x nx), B (nx x nu) */
    double rho, umax;
} Family;

static TinyMatrixPOD mat(double* d, int r, int c) { TinyMatrixPOD m; m.data = d; m.rows = r; m.cols = c; return m; }

/* family 0: a chain of two double integrators (nx = 4, nu = 2); family 1: a damped oscillator pair with cross coupling (nx = 6, nu = 3) */
static void make_family(Family* f, int which) {
    memset(f, 0, sizeof(*f));
    const double dt = 0.05;
    if (which == 0) {
        f->nx = 4; f->nu = 2; f->N = 10; f->rho = 1.0; f->umax = 0.4;
        for (int i = 0; i < 4; ++i) f->A[i + 4 * i] = 1.0;
        f->A[0 + 4 * 2] = dt; f->A[1 + 4 * 3] = dt;
        f->B[0 + 4 * 0] = 0.5 * dt * dt; f->B[2 + 4 * 0] = dt; f->B[1 + 4 * 1] = 0.5 * dt * dt; f->B[3 + 4 * 1] = dt;
        for (int i = 0; i < 4; ++i) f->Q[i] = i < 2 ? 10.0 : 1.0;
        for (int i = 0; i < 2; ++i) f->R[i] = 0.5;
    } else {
        f->nx = 6; f->nu = 3; f->N = 10; f->rho = 2.0; f->umax = 0.6;
        for (int i = 0; i < 6; ++i) f->A[i + 6 * i] = 1.0;
        for (int i = 0; i < 3; ++i) {
            f->A[i + 6 * (3 + i)] = dt;                       /* position <- velocity */
            f->A[(3 + i) + 6 * i] = -0.8 * dt;                /* spring */
            f->A[(3 + i) + 6 * (3 + i)] = 1.0 - 0.1 * dt;     /* damping */
            f->A[(3 + i) + 6 * ((i + 1) % 3)] += 0.2 * dt;    /* coupling */
            f->B[(3 + i) + 6 * i] = dt;
        }
        for (int i = 0; i < 6; ++i) f->Q[i] = i < 3 ? 8.0 : 0.5;
        for (int i = 0; i < 3; ++i) f->R[i] = 0.3;
    }
}

typedef struct {
    Family fam;
    TinySolver* solver;
    double x[NMAX], x_start[NMAX];
    int steps;
    long iters;
    int unsolved, rc;
    double seconds;
} Job;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int setup_job(Job* j, int which, int id, int steps) {
    make_family(&j->fam, which);
    Family* f = &j->fam;
    const int nx = f->nx, nu = f->nu, N = f->N;
    double Qd[NMAX * NMAX] = {0}, Rd[NMAX * NMAX] = {0}, fd[NMAX] = {0};
    for (int i = 0; i < nx; ++i) Qd[i + nx * i] = f->Q[i];
    for (int i = 0; i < nu; ++i) Rd[i + nu * i] = f->R[i];
    TinyMatrixPOD A = mat(f->A, nx, nx), B = mat(f->B, nx, nu), fm = mat(fd, nx, 1), Q = mat(Qd, nx, nx), R = mat(Rd, nu, nu);
    if (tiny_setup(&j->solver, &A, &B, &fm, &Q, &R, f->rho, nx, nu, N, 0)) return 1;
    double* xmin = malloc(sizeof(double) * nx * N), * xmax = malloc(sizeof(double) * nx * N);
    double* umin = malloc(sizeof(double) * nu * (N - 1)), * umax = malloc(sizeof(double) * nu * (N - 1));
    for (int i = 0; i < nx * N; ++i) { xmin[i] = -10.0; xmax[i] = 10.0; }
    for (int i = 0; i < nu * (N - 1); ++i) { umin[i] = -f->umax; umax[i] = f->umax; }
    TinyMatrixPOD m0 = mat(xmin, nx, N), m1 = mat(xmax, nx, N), m2 = mat(umin, nu, N - 1), m3 = mat(umax, nu, N - 1);
    const int rc = tiny_set_bound_constraints(j->solver, &m0, &m1, &m2, &m3);
    free(xmin); free(xmax); free(umin); free(umax);
    if (rc) return 2;
    j->solver->settings->max_iter = 60;
    for (int i = 0; i < nx; ++i) j->x_start[i] = (i < nx / 2 ? 1.0 : 0.2) * (1.0 + 0.13 * id) * ((i & 1) ? -1.0 : 1.0);
    j->steps = steps;
    return 0;
}

/* one closed-loop episode from the cold state: tiny_set_x0 -> tiny_solve -> x <- A x + B u[:,0] (examples/quadrotor_hovering.cpp:73-93) */
static void* episode(void* arg) {
    Job* j = (Job*)arg;
    Family* f = &j->fam;
    const int nx = f->nx, nu = f->nu, N = f->N;
    TinyWorkspace* w = j->solver->work;
    memcpy(j->x, j->x_start, sizeof(double) * nx);
    /* the state tiny_setup leaves behind: every warm-start field zero */
    TinyMatrixPOD* zero[] = {&w->x, &w->u, &w->v, &w->vnew, &w->z, &w->znew, &w->g, &w->y};
    for (unsigned k = 0; k < sizeof(zero) / sizeof(zero[0]); ++k) memset(zero[k]->data, 0, sizeof(double) * zero[k]->rows * zero[k]->cols);
    (void)N;
    j->iters = 0; j->unsolved = 0; j->rc = 0;
    const double t0 = now();
    for (int k = 0; k < j->steps; ++k) {
        TinyVectorPOD x0; x0.data = j->x; x0.rows = nx;
        if (tiny_set_x0(j->solver, &x0)) { j->rc = 3; break; }
        const int r = tiny_solve(j->solver);
        if (r != 0 && r != 1) { j->rc = 100 + r; break; }
        j->unsolved += r;
        j->iters += j->solver->solution->iter;
        double xn[NMAX];
        for (int i = 0; i < nx; ++i) {
            double a = 0.0;
            for (int c = 0; c < nx; ++c) a += f->A[i + nx * c] * j->x[c];
            for (int c = 0; c < nu; ++c) a += f->B[i + nx * c] * w->u.data[c];      /* u[:,0] */
            xn[i] = a;
        }
        memcpy(j->x, xn, sizeof(double) * nx);
    }
    j->seconds = now() - t0;
    return NULL;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8, steps = argc > 2 ? atoi(argv[2]) : 200;
    if (T < 1 || T > 64 || steps < 1) { fprintf(stderr, "usage: dropin_threads [threads 1..64] [steps]\n"); return 2; }
    if (tiny_batch_device_count() <= 0) { fprintf(stderr, "no GPU: libtinympc_amd has no CPU path\n"); return 3; }
    Job* jobs = calloc(T, sizeof(Job));
    for (int t = 0; t < T; ++t)
        if (setup_job(&jobs[t], t & 1, t, steps)) { fprintf(stderr, "setup of solver %d failed\n", t); return 4; }
    /* serial pass (also pays every first-use cost: device contexts, tables) */
    long ser_iters[64]; double ser_x[64][NMAX]; int ser_uns[64];
    episode(&jobs[0]);                                    /* warm-up of the library itself */
    double t0 = now();
    for (int t = 0; t < T; ++t) {
        episode(&jobs[t]);
        if (jobs[t].rc) { fprintf(stderr, "solver %d: error %d in the serial pass\n", t, jobs[t].rc); return 5; }
        ser_iters[t] = jobs[t].iters; ser_uns[t] = jobs[t].unsolved; memcpy(ser_x[t], jobs[t].x, sizeof(double) * NMAX);
    }
    const double serial_s = now() - t0;
    /* concurrent pass */
    pthread_t th[64];
    t0 = now();
    for (int t = 0; t < T; ++t) pthread_create(&th[t], NULL, episode, &jobs[t]);
    for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
    const double conc_s = now() - t0;
    int bad = 0;
    for (int t = 0; t < T; ++t) {
        const int same = jobs[t].rc == 0 && jobs[t].iters == ser_iters[t] && jobs[t].unsolved == ser_uns[t] &&
                         memcmp(jobs[t].x, ser_x[t], sizeof(double) * jobs[t].fam.nx) == 0;
        printf("solver %d (nx %d nu %d): %ld iterations over %d steps, %d at max_iter, |x_final| %.3e  %s\n", t, jobs[t].fam.nx, jobs[t].fam.nu,
               jobs[t].iters, steps, jobs[t].unsolved, fabs(jobs[t].x[0]) + fabs(jobs[t].x[1]), same ? "== serial" : "DIFFERS from the serial pass");
        bad += !same;
    }
    printf("threads %d steps %d: serial %.1f us per tiny_solve, concurrent %.1f us per tiny_solve (wall / total solves), throughput ratio %.2f\n",
           T, steps, 1e6 * serial_s / ((double)T * steps), 1e6 * conc_s / ((double)T * steps), serial_s / conc_s);
    for (int t = 0; t < T; ++t) tiny_destroy(jobs[t].solver);
    free(jobs);
    if (bad) { printf("FAILED: %d solver(s) differ\n", bad); return 1; }
    printf("OK\n");
    return 0;
}
