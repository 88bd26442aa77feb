/* Plain-C consumer of libhipdrt.so: what a non-Python binding (cgo / JNI / Fortran ...) of the hybrid-drt hot path
 * would look like.  Solves two small box-constrained QPs (cvxopt.solvers.qp(P, q, G = -I, h) semantics) and builds a
 * 4 x 6 impedance matrix in trapz mode; prints everything as text for tests/test_gpu_cabi_c.py to compare with the
 * CPU checker.  Build:  gcc cabi_example.c -I../../include -L../../hybrid-drt_amd -lhipdrt -o cabi_example        */
#include <stdio.h>
#include <stdlib.h>
#include "hipdrt.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hipdrt_last_error()); return 1; } } while (0)

int main(void) {
    hipdrt_ctx* ctx = NULL;
    CHECK(hipdrt_create(0, &ctx));
    char arch[64]; int ncu = 0; long long hbm = 0;
    CHECK(hipdrt_device_info(ctx, arch, sizeof arch, &ncu, &hbm));
    printf("arch %s\n", arch);

    /* two 3x3 problems: minimise 1/2 x'Px + q'x  s.t.  x >= -h */
    enum { B = 2, N = 3 };
    const double P[B][N][N] = {{{4, 1, 0}, {1, 3, 1}, {0, 1, 2}}, {{2, 0, 0}, {0, 2, 0}, {0, 0, 2}}};
    const double q[B][N] = {{1, -2, 1}, {-1, 1, -3}};
    const double h[B][N] = {{0, 0, 0}, {0, 0, 0}};
    double x[B][N], pcost[B];
    int iters[B], status[B];
    CHECK(hipdrt_qp_batch(ctx, B, N, 1, &P[0][0][0], &q[0][0], 1, &h[0][0], NULL, &x[0][0], iters, pcost, status));
    for (int b = 0; b < B; ++b)
        printf("qp %d status %d iters %d x %.17g %.17g %.17g pcost %.17g\n", b, status[b], iters[b], x[b][0], x[b][1],
               x[b][2], pcost[b]);

    /* Z', Z'' for 4 frequencies x 6 time constants, 1000-point trapezoid (mode 1 = TRAPZ) */
    const double freq[4] = {1e3, 1e2, 1e1, 1e0};
    const double tau[6] = {1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1};
    double are[4][6], aim[4][6];
    CHECK(hipdrt_impedance_matrix(ctx, 1, 0, freq, 4, tau, 6, HIPDRT_MODE_TRAPZ, 0, 0.43429448190325176, 0, NULL, NULL, NULL,
                                  NULL, 1000, &are[0][0], &aim[0][0]));
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 6; ++c) printf("a %d %d %.17g %.17g\n", r, c, are[r][c], aim[r][c]);
    /* round 6: the potentiostatic delta response (two voltage steps, 3 samples x 2 time constants) and what a staged
     * 256 x 512 spectrum costs a plan */
    const double times[3] = {0.5, 1.5, 3.0}, ptau[2] = {0.5, 2.0}, st[2] = {1.0, 2.0}, sa[2] = {1e-3, -2e-3};
    double pa[3][2];
    CHECK(hipdrt_response_matrix_variant(ctx, times, 3, ptau, 2, st, sa, NULL, 2, HIPDRT_RESPONSE_POT, 0.0, 0, &pa[0][0], NULL));
    for (int r = 0; r < 3; ++r) printf("pot %d %.17g %.17g\n", r, pa[r][0], pa[r][1]);
    long long per = 0;
    CHECK(hipdrt_plan_bytes_per_spectrum(256, 512, 2, &per));
    printf("bytes_per_spectrum %lld\n", per);
    CHECK(hipdrt_destroy(ctx));
    return 0;
}
