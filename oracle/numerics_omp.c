/* C / OpenMP restatement of the reference's numba-jitted inner functions -- TEST INFRASTRUCTURE.
 *
 * See oracle/__init__.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this.  Every function follows the function of the same name in
 * /root/reference/src/vilma/numerics.py (line ranges cited) loop for loop: the reference's
 * `for i in prange(N)` (numba parallel=True: the SNP loop spread over all cores) is
 * `#pragma omp parallel for` over the same index with the same loop nest inside, so timing these
 * on the GPU box's host stands in for the reference's compiled CPU path (bench.py cpu_baseline,
 * kind "port").  Layouts are the reference's: vi_mu [M,P,N], vi_delta [N,M], vi_sigma /
 * nat_sigma [M,P,P,N], hyper_delta [A,M]; all C-contiguous float64, annotations int64.
 *
 * Parity: PINNED by tests/test_oracle_native.py against tests/golden/numerics_kat.npz (vectors
 * produced by the reference itself) and against the numpy restatement oracle/numerics.py.
 *
 * Build (oracle/Makefile):  gcc -O3 -fopenmp -fPIC -shared -o oracle/_build/libnumerics_omp.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define EPSILON 1e-100 /* numerics.py:8 */

/* numerics.py:11-15 */
void sum_betas(const double *old_beta, const double *new_beta, double step, int64_t n,
               double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; ++t) out[t] = step * new_beta[t] + (1. - step) * old_beta[t];
}

/* numerics.py:18-21 */
void fast_divide(const double *x, const double *y, int64_t n, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; ++t) out[t] = x[t] / y[t];
}

/* numerics.py:24-28: w / x - y * z */
void fast_linked_ests(const double *w, const double *x, const double *y, const double *z,
                      int64_t n, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; ++t) out[t] = w[t] / x[t] - y[t] * z[t];
}

/* numerics.py:31-46; all [P,N] */
double fast_likelihood(const double *post_means, const double *post_vars, const double *scaled_mu,
                       const double *scaled_ld_diags, const double *linked_ests,
                       const double *adj_marginal, const double *chi_stat,
                       const double *ld_ranks, const double *error_scaling, int P, int64_t N) {
    double total = 0.0;
    for (int p = 0; p < P; ++p) {
        double lik = 0.0;
        const int64_t o = (int64_t)p * N;
#pragma omp parallel for schedule(static) reduction(+ : lik)
        for (int64_t i = 0; i < N; ++i)
            lik += -0.5 * (scaled_ld_diags[o + i] * post_vars[o + i]
                           + linked_ests[o + i] * scaled_mu[o + i])
                   + post_means[o + i] * adj_marginal[o + i];
        lik += -0.5 * chi_stat[p];
        total += lik / error_scaling[p] - 0.5 * ld_ranks[p] * log(error_scaling[p]);
    }
    return total;
}

/* numerics.py:49-57: out[p,i] = sum_k vi_mu[k,p,i] vi_delta[i,k] */
void fast_posterior_mean(const double *vi_mu, const double *vi_delta, int M, int P, int64_t N,
                         double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        for (int p = 0; p < P; ++p) {
            double s = 0.0;
            for (int k = 0; k < M; ++k) s += vi_mu[((int64_t)k * P + p) * N + i] * vi_delta[i * M + k];
            out[(int64_t)p * N + i] = s;
        }
}

/* numerics.py:60-65: second moment (temp + vi_mu^2 materialised by the reference) minus mean^2 */
void fast_pmv(const double *mean, const double *vi_mu, const double *vi_delta, const double *temp,
              int M, int P, int64_t N, double *out) {
    const int64_t n = (int64_t)M * P * N;
    double *second = (double *)malloc((size_t)n * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; ++t) second[t] = temp[t] + vi_mu[t] * vi_mu[t];
    fast_posterior_mean(second, vi_delta, M, P, N, out);
    free(second);
    const int64_t pn = (int64_t)P * N;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < pn; ++t) out[t] -= mean[t] * mean[t];
}

/* numerics.py:68-95: out[s,p,i] = scale * sum_q nat_sigma[s,p,q,i] vi_mu[s,q,i] */
static void nat_inner(const double *vi_mu, const double *nat_sigma, int M, int P, int64_t N,
                      double scale, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        for (int p = 0; p < P; ++p)
            for (int s = 0; s < M; ++s) {
                double t = 0.0;
                for (int q = 0; q < P; ++q)
                    t += nat_sigma[(((int64_t)s * P + p) * P + q) * N + i]
                         * vi_mu[((int64_t)s * P + q) * N + i];
                out[((int64_t)s * P + p) * N + i] = scale * t;
            }
}
void fast_nat_inner_product_m2(const double *vi_mu, const double *nat_sigma, int M, int P,
                               int64_t N, double *out) {
    nat_inner(vi_mu, nat_sigma, M, P, N, -2.0, out);
}
void fast_nat_inner_product(const double *vi_mu, const double *nat_sigma, int M, int P, int64_t N,
                            double *out) {
    nat_inner(vi_mu, nat_sigma, M, P, N, 1.0, out);
}

/* numerics.py:98-115; mixture_prec [M,P,P] (the trailing unit axis dropped) */
double fast_inner_product_comp(const double *vi_mu, const double *mixture_prec,
                               const double *vi_delta, int M, int P, int64_t N) {
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t i = 0; i < N; ++i)
        for (int k = 0; k < M; ++k) {
            double t = 0.0;
            for (int p = 0; p < P; ++p)
                for (int q = 0; q < P; ++q)
                    t += vi_mu[((int64_t)k * P + p) * N + i] * vi_mu[((int64_t)k * P + q) * N + i]
                         * mixture_prec[((int64_t)k * P + q) * P + p];
            total += t * vi_delta[i * M + k];
        }
    return 0.5 * total;
}

/* numerics.py:118-129 */
void sum_annotations(const double *deltas, const int64_t *annotations, int A, int M, int64_t N,
                     double *out /*[A,M]*/) {
    for (int a = 0; a < A; ++a) {
#pragma omp parallel
        {
            double *part = (double *)calloc((size_t)M, sizeof(double));
#pragma omp for schedule(static) nowait
            for (int64_t i = 0; i < N; ++i)
                if (annotations[i] == a)
                    for (int k = 0; k < M; ++k) part[k] += deltas[i * M + k];
#pragma omp critical
            for (int k = 0; k < M; ++k) out[(int64_t)a * M + k] += part[k];
            free(part);
        }
    }
}

/* numerics.py:132-141 */
double fast_delta_kl(const double *vi_delta, const double *hyper_delta, const int64_t *annotations,
                     int A, int M, int64_t N) {
    double *log_hyper = (double *)malloc((size_t)A * M * sizeof(double));
    for (int t = 0; t < A * M; ++t) log_hyper[t] = log(hyper_delta[t]);
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t i = 0; i < N; ++i) {
        const double *lh = log_hyper + annotations[i] * M;
        double s = 0.0;
        for (int k = 0; k < M; ++k) s += vi_delta[i * M + k] * (log(vi_delta[i * M + k]) - lh[k]);
        total += s;
    }
    free(log_hyper);
    return total;
}

/* numerics.py:144-146 */
double fast_beta_kl(const double *sigma_summary, const double *vi_delta, int64_t n) {
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t t = 0; t < n; ++t) total += sigma_summary[t] * vi_delta[t];
    return 0.5 * total;
}

/* numerics.py:149-164; out [N, M-1] */
void fast_vi_delta_grad(const double *hyper_delta, const double *log_det,
                        const int64_t *annotations, int A, int M, int64_t N, double *out) {
    double *full = (double *)malloc((size_t)A * M * sizeof(double));
    for (int a = 0; a < A; ++a)
        for (int k = 0; k < M; ++k)
            full[a * M + k] = log(hyper_delta[a * M + k]) + -0.5 * log_det[k];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const double *row = full + annotations[i] * M;
        const double last = row[M - 1];
        for (int k = 0; k < M - 1; ++k) out[i * (M - 1) + k] = row[k] - last;
    }
    free(full);
}

/* numerics.py:179-195 applied to one row of K logits; out has K + 1 entries */
static void invert_row(const double *probs, int K, double *out) {
    double max_p = 0.0;                       /* np.maximum(np.max(probs[i]), 0) */
    for (int k = 0; k < K; ++k) max_p = probs[k] > max_p ? probs[k] : max_p;
    const double last_p = exp(-max_p);
    double denom = last_p;
    for (int k = 0; k < K; ++k) {
        out[k] = exp(probs[k] - max_p);
        denom += out[k];
    }
    for (int k = 0; k < K; ++k) out[k] = fmax(out[k] / denom, EPSILON);
    out[K] = fmax(last_p / denom, EPSILON);
}

/* numerics.py:198-213; const_part [N,M], nat_vi_delta [N,M-1], out [N,M] */
void fast_invert_nat_vi_delta(const double *new_mu, const double *nat_mu, const double *const_part,
                              const double *nat_vi_delta, int M, int P, int64_t N, double *out) {
#pragma omp parallel
    {
        double *to_invert = (double *)malloc((size_t)M * sizeof(double));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            double last = const_part[i * M + (M - 1)];
            for (int j = 0; j < P; ++j)
                last += new_mu[((int64_t)(M - 1) * P + j) * N + i] * nat_mu[((int64_t)(M - 1) * P + j) * N + i];
            for (int k = 0; k < M - 1; ++k) {
                double add = const_part[i * M + k];
                for (int j = 0; j < P; ++j)
                    add += new_mu[((int64_t)k * P + j) * N + i] * nat_mu[((int64_t)k * P + j) * N + i];
                to_invert[k] = 0.5 * (add - last) + nat_vi_delta[i * (M - 1) + k];
            }
            invert_row(to_invert, M - 1, out + i * M);
        }
        free(to_invert);
    }
}
