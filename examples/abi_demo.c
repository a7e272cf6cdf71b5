/* abi_demo.c -- the C ABI of include/acme_hip.h used from plain C, no Python, no C++.
 *
 * Runs BASELINE config 1's circuit (examples/diodeclipper.jl at 44.1 kHz, one second of a 1 kHz
 * sine) as a batch of 4 amplitudes on the GPU and checks the unit-amplitude instance against the
 * output the reference's documentation prints (docs/src/gettingstarted.md:106-113).
 *
 *   gcc -O2 -Iinclude examples/abi_demo.c -o examples/abi_demo \
 *       -Lacme_jl_amd/csrc -lacme_hip -Wl,-rpath,'$ORIGIN/../acme_jl_amd/csrc' \
 *       -Wl,-rpath-link,/opt/rocm/lib -lm
 *
 * The matrices are the DiscreteModel fields ACME derives for this circuit (a Julia binding passes
 * model.a, model.b, ... directly), column-major.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "acme_hip.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != ACME_OK) {                                                  \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, acme_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(void) {
    /* nx = 1, nu = 1, ny = 1; one nonlinear sub-problem {d1, d2}: nn = 2, nq = 4, np = 1 */
    const double a[1] = {-1.0}, b[1] = {0.0}, c[2] = {9.4e-08, 0.0}, x0[1] = {0.0};
    const double dy[1] = {0.0}, ey[1] = {0.0}, fy[2] = {1.0, 0.0}, y0[1] = {0.0};
    const double pexp[4] = {0.0, 1.0, 0.0, 0.0};                           /* 4 x 1 */
    const double dq[1] = {88200.0}, eq[1] = {0.001}, fqprev[2] = {0.0, 0.0};
    const double fq[8] = {1.0, -0.0051454, -1.0, 0.0, 0.0, 1.0, 0.0, 1.0};   /* 4 x 2, column-major */
    const double q0[4] = {0.0, 0.0, 0.0, 0.0}, init_z[2] = {0.0, 0.0};
    /* element table in CircuitNLFunc order: d1 = diode(is=1e-15), d2 = diode(is=1.8e-15) */
    const int kind[2] = {ACME_KIND_DIODE, ACME_KIND_DIODE}, qoff[2] = {0, 2}, roff[2] = {0, 1};
    double par[2 * ACME_MAX_ELEM_PAR] = {0};
    par[0] = 1e-15;                      par[1] = 1.0;
    par[ACME_MAX_ELEM_PAR] = 1.8e-15;    par[ACME_MAX_ELEM_PAR + 1] = 1.0;

    enum { N = 4, T = 44100 };
    const double amp[N] = {0.1, 1.0, 3.0, 10.0};
    const double pi = 3.14159265358979323846;

    acme_model *m = NULL;
    acme_batch *bt = NULL;
    acme_options o;
    CHECK(acme_model_create(1, 1, 1, 2, a, b, c, x0, dy, ey, fy, y0, &m));
    CHECK(acme_model_add_subproblem(m, 2, 4, 1, pexp, dq, eq, fqprev, fq, q0, init_z, 2, kind, qoff, roff, par));
    acme_default_options(&o);
    o.solver = ACME_SOLVER_CACHING_HOMOTOPY;     /* the reference's default stack */
    CHECK(acme_batch_create(m, N, &o, &bt));

    double *u = malloc(sizeof(double) * N * T), *y = malloc(sizeof(double) * N * T);
    if (!u || !y) return 1;
    for (int i = 0; i < N; ++i)
        for (int n = 0; n < T; ++n) u[(size_t)i * T + n] = amp[i] * sin(2 * pi * 1000.0 / 44100.0 * n);
    CHECK(acme_batch_run(bt, u, y, T, ACME_MEM_HOST, NULL));

    acme_report rep[N];
    CHECK(acme_batch_get_report(bt, rep));
    for (int i = 0; i < N; ++i) {
        printf("amplitude %5.1f V: y[1..3] = %.6g %.6g %.6g   Newton iterations/sample %.2f, warnings %lld\n",
               amp[i], y[(size_t)i * T + 1], y[(size_t)i * T + 2], y[(size_t)i * T + 3],
               (double)rep[i].iters_total / T, rep[i].n_warn);
        if (rep[i].n_warn != 0 || rep[i].first_nonfinite >= 0) return 2;
    }
    /* docs/src/gettingstarted.md:106-113, printed to 6 significant digits */
    const double head[4] = {0.0, 0.0275964, 0.0990996, 0.195777};
    const double tail[3] = {-0.537508, -0.462978, -0.36521};
    const double *y1 = y + (size_t)1 * T;
    int bad = 0;
    for (int k = 0; k < 4; ++k) bad |= fabs(y1[k] - head[k]) > 5e-6 * fmax(1e-1, fabs(head[k]));
    for (int k = 0; k < 3; ++k) bad |= fabs(y1[T - 3 + k] - tail[k]) > 5e-6;
    /* the solver plugin surface: get_extrapolation_jacobian = dz/dp at each instance's origin
     * (nn x np = 2 x 1 here), finite for every instance after a converged run */
    double jac[N * 2 * 1];
    CHECK(acme_batch_get_extrapolation_jacobian(bt, 0, jac, ACME_MEM_HOST, NULL));
    for (int i = 0; i < N; ++i) {
        printf("amplitude %5.1f V: dz/dp at the origin = [%.6g, %.6g]\n", amp[i], jac[2 * i], jac[2 * i + 1]);
        if (!(jac[2 * i] == jac[2 * i]) || !(jac[2 * i + 1] == jac[2 * i + 1])) return 4;
    }
    float ms = 0.f;
    CHECK(acme_batch_last_kernel_ms(bt, &ms));
    printf("doctest vector %s; last kernel %.3f ms\n", bad ? "MISMATCH" : "reproduced", ms);
    free(u);
    free(y);
    acme_batch_destroy(bt);
    acme_model_destroy(m);
    return bad ? 3 : 0;
}
