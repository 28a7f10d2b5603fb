/* A complete fit driven from plain C through include/vilma_hip.h: what a non-Python host (the
 * cgo / JNI / Rust-FFI stub of INTEGRATION.md) does, with no Python, torch or HIP header in sight.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/host_fit.c -o host_fit \
 *       -Lvilma_amd -l:libvilma_hip.so -Wl,-rpath,$PWD/vilma_amd -lm
 *   ./host_fit problem.bin [n_sweeps] [lookahead]
 *
 * The loop is optimize()'s (/root/reference/src/vilma/variational_inference.py:353-389): L = ones(5),
 * line_search_rate 2, stop when no posterior mean moved.  Per sweep one line goes to stdout:
 *   sweep <it> elbo <%.17g> L <5 x %.17g> evaluations <n> moved <count>
 * and at the end `posterior_checksum <sum of means> <sum of variances>`.
 *
 * problem.bin (little-endian, written by tests/test_gpu_c_host.py from a golden trajectory):
 *   int64  P, N, M, A, scale_se, n_ld
 *   double adj[P*N], se[P*N], sld[P*N], scalings[P*N];  int32 annot[N]
 *   double prec[M*P*P], log_det[M], counts[A], chi[P], ranks[P], fake_mu[P*N];  int64 perm[N]
 *   per cohort: int64 n_blocks; per block: int64 n; double R[n*n]   (dense, row-major)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "vilma_hip.h"

static FILE *in;

static void need(void *dst, size_t size, size_t count) {
    if (fread(dst, size, count, in) != count) {
        fprintf(stderr, "host_fit: short read\n");
        exit(2);
    }
}
static double *doubles(size_t n) {
    double *p = (double *)malloc((n ? n : 1) * sizeof(double));
    if (!p) exit(3);
    need(p, sizeof(double), n);
    return p;
}
static vilma_ctx *ctx;
static void ok(int rc, const char *what) {
    if (rc != 0) {
        fprintf(stderr, "host_fit: %s: %s\n", what, vilma_last_error(ctx));
        exit(1);
    }
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s problem.bin [n_sweeps] [lookahead]\n", argv[0]);
        return 2;
    }
    in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 2; }
    const int n_sweeps = argc > 2 ? atoi(argv[2]) : 20;
    const int lookahead = argc > 3 ? atoi(argv[3]) : 1;

    int64_t h[6];
    need(h, sizeof(int64_t), 6);
    const int P = (int)h[0], M = (int)h[2], A = (int)h[3], scale_se = (int)h[4];
    const int64_t N = h[1], n_ld = h[5];
    const size_t PN = (size_t)P * (size_t)N;

    ok(vilma_create(P, N, M, A, &ctx), "vilma_create");
    double *adj = doubles(PN), *se = doubles(PN), *sld = doubles(PN), *scal = doubles(PN);
    int32_t *annot = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    need(annot, sizeof(int32_t), (size_t)N);
    ok(vilma_set_snp_data(ctx, adj, se, sld, scal, annot), "vilma_set_snp_data");
    double *prec = doubles((size_t)M * P * P), *log_det = doubles((size_t)M);
    ok(vilma_set_mixture(ctx, prec, log_det), "vilma_set_mixture");
    double *counts = doubles((size_t)A);
    ok(vilma_set_annotation_counts(ctx, counts), "vilma_set_annotation_counts");
    double *chi = doubles((size_t)P), *ranks = doubles((size_t)P), *fake_mu = doubles(PN);
    int64_t *perm = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    need(perm, sizeof(int64_t), (size_t)N);

    for (int p = 0; p < P; ++p) {
        int64_t n_blocks;
        need(&n_blocks, sizeof(int64_t), 1);
        /* two passes over the cohort's records: sizes first (vilma_ld_begin wants the total) */
        const long at = ftell(in);
        int64_t total = 0;
        for (int64_t b = 0; b < n_blocks; ++b) {
            int64_t n;
            need(&n, sizeof(int64_t), 1);
            total += vilma_ld_dense_elems((int)n);
            fseek(in, (long)(n * n * (int64_t)sizeof(double)), SEEK_CUR);
        }
        fseek(in, at, SEEK_SET);
        ok(vilma_ld_begin(ctx, p, (int)n_blocks, n_ld, perm, total), "vilma_ld_begin");
        for (int64_t b = 0; b < n_blocks; ++b) {
            int64_t n;
            need(&n, sizeof(int64_t), 1);
            double *R = doubles((size_t)(n * n));
            ok(vilma_ld_add_dense(ctx, p, (int)n, R), "vilma_ld_add_dense");
            free(R);
        }
        ok(vilma_ld_end(ctx, p), "vilma_ld_end");
    }
    fclose(in);
    ok(vilma_set_fit_constants(ctx, chi, ranks, scale_se), "vilma_set_fit_constants");

    double elbo;
    ok(vilma_initialize(ctx, NULL, fake_mu, &elbo), "vilma_initialize");
    printf("init elbo %.17g\n", elbo);
    ok(vilma_snapshot_mean(ctx, NULL), "vilma_snapshot_mean");

    double L[5] = {1.0, 1.0, 1.0, 1.0, 1.0};
    double running = NAN;                      /* None: first sweep */
    vilma_sweep_stats st;
    for (int it = 0; it < n_sweeps; ++it) {
        int flags = VILMA_SWEEP_DIFF;
        /* promise the next call (and let the device veto it when nothing moved) except on the last */
        if (lookahead && it + 1 < n_sweeps) flags |= VILMA_SWEEP_LOOKAHEAD | VILMA_SWEEP_VETO;
        ok(vilma_sweep(ctx, NULL, L, &elbo, &running, 2.0, flags, &st), "vilma_sweep");
        printf("sweep %d elbo %.17g L %.17g %.17g %.17g %.17g %.17g evaluations %d moved %.0f\n", it,
               elbo, L[0], L[1], L[2], L[3], L[4], (int)st.n_evaluations, st.diff_sum[0]);
        if (st.diff_sum[0] == 0.0) break;      /* optimize()'s stopping rule */
    }
    ok(vilma_sweep_drain(ctx), "vilma_sweep_drain");

    double *mean = (double *)malloc(PN * sizeof(double)), *var = (double *)malloc(PN * sizeof(double));
    ok(vilma_posterior(ctx, mean, var), "vilma_posterior");
    double sm = 0.0, sv = 0.0;
    for (size_t i = 0; i < PN; ++i) { sm += mean[i]; sv += var[i]; }
    printf("posterior_checksum %.17g %.17g\n", sm, sv);
    double check;
    ok(vilma_elbo(ctx, &check), "vilma_elbo");
    printf("final elbo %.17g\n", check);
    vilma_destroy(ctx);
    return 0;
}
