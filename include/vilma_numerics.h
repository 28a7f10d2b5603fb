/* vilma_numerics.h -- C-ABI of the Function API (part of libvilma_hip.so).
 *
 * The reference's third boundary depth: the 20 pure functions of
 * /root/reference/src/vilma/numerics.py.  Each entry point below names the reference function it
 * replaces (file:line); arrays are float64 / int64 in the reference's C-order layouts
 *     vi_mu [M,P,N]   vi_delta [N,M]   vi_sigma, nat_sigma [M,P,P,N]   hyper_delta [A,M]
 * and may live in host or device memory (the library copies with hipMemcpyDefault).  `out` is
 * caller-allocated with the reference's return shape.  Calls are synchronous and stateless;
 * 0 = ok, otherwise vilma_num_last_error() (thread-local) holds the message.  The arithmetic runs
 * on the GPU only: without a HIP device every call fails, there is no host path.
 *
 * The fit does not use these (its per-SNP work is fused, see vilma_hip.h); they exist so that
 * code written against `vilma.numerics` keeps working.  Python face: vilma_amd/numerics.py.
 */
#ifndef VILMA_NUMERICS_H
#define VILMA_NUMERICS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *vilma_num_last_error(void);

/* sum_betas (numerics.py:11-15): out = step*new + (1-step)*old over n elements */
int vilma_num_sum_betas(const double *old_beta, const double *new_beta, double step_size,
                        int64_t n, double *out);
/* fast_divide (numerics.py:18-21) */
int vilma_num_divide(const double *x, const double *y, int64_t n, double *out);
/* fast_linked_ests (numerics.py:24-28): w/x - y*z */
int vilma_num_linked_ests(const double *w, const double *x, const double *y, const double *z,
                          int64_t n, double *out);
/* fast_likelihood (numerics.py:31-46): six [P,N] arrays, three [P] -> out[1] */
int vilma_num_likelihood(const double *post_means, const double *post_vars,
                         const double *scaled_mu, const double *scaled_ld_diags,
                         const double *linked_ests, const double *adj_marginal,
                         const double *chi_stat, const double *ld_ranks,
                         const double *error_scaling, int P, int64_t N, double *out);
/* fast_posterior_mean (numerics.py:49-57): out [P,N] */
int vilma_num_posterior_mean(const double *vi_mu, const double *vi_delta, int M, int P, int64_t N,
                             double *out);
/* fast_pmv (numerics.py:60-65): mean [P,N], temp [M,P,N] -> out [P,N] */
int vilma_num_pmv(const double *mean, const double *vi_mu, const double *vi_delta,
                  const double *temp, int M, int P, int64_t N, double *out);
/* fast_nat_inner_product (scale 1, numerics.py:83-95) and fast_nat_inner_product_m2 (scale -2,
 * numerics.py:68-80): out [M,P,N] */
int vilma_num_nat_inner_product(const double *vi_mu, const double *nat_sigma, int M, int P,
                                int64_t N, double scale, double *out);
/* fast_inner_product_comp (numerics.py:98-115): mixture_prec [M,P,P] -> out[1] */
int vilma_num_inner_product_comp(const double *vi_mu, const double *mixture_prec,
                                 const double *vi_delta, int M, int P, int64_t N, double *out);
/* sum_annotations (numerics.py:118-129): deltas [N,M], annotations [N] -> out [A,M] */
int vilma_num_sum_annotations(const double *deltas, const int64_t *annotations, int A, int M,
                              int64_t N, double *out);
/* fast_delta_kl (numerics.py:132-141) -> out[1] */
int vilma_num_delta_kl(const double *vi_delta, const double *hyper_delta,
                       const int64_t *annotations, int A, int M, int64_t N, double *out);
/* fast_beta_kl (numerics.py:144-146): 0.5 * sum(sigma_summary * vi_delta) -> out[1] */
int vilma_num_beta_kl(const double *sigma_summary, const double *vi_delta, int64_t n, double *out);
/* fast_vi_delta_grad (numerics.py:149-164): out [N, M-1] */
int vilma_num_vi_delta_grad(const double *hyper_delta, const double *log_det,
                            const int64_t *annotations, int A, int M, int64_t N, double *out);
/* map_to_nat_cat_2D (numerics.py:167-176): probs [N,K] -> out [N,K-1] */
int vilma_num_map_to_nat_cat(const double *probs, int64_t N, int K, double *out);
/* invert_nat_cat_2D (numerics.py:179-195): probs [N,K] -> out [N,K+1], clamped at 1e-100 */
int vilma_num_invert_nat_cat(const double *probs, int64_t N, int K, double *out);
/* fast_invert_nat_vi_delta (numerics.py:198-213): new_mu, nat_mu [M,P,N], const_part [N,M],
 * nat_vi_delta [N,M-1] -> out [N,M] */
int vilma_num_invert_nat_vi_delta(const double *new_mu, const double *nat_mu,
                                  const double *const_part, const double *nat_vi_delta, int M,
                                  int P, int64_t N, double *out);
/* matrix_invert / _matrix_invert_4d_numba (numerics.py:216-244): mats [n,P,P], P <= 4.
 * closed_form != 0: the reference's 4-D special case for P <= 2 (2 x 2 written symmetric);
 * otherwise pivoted elimination (np.linalg.inv). */
int vilma_num_matrix_invert(const double *mats, int64_t n, int P, int closed_form, double *out);
/* matrix_log_det / _matrix_log_det_4d_numba (numerics.py:257-280): out [n]; closed_form as above
 * (log of the signed determinant), otherwise log|det| (np.linalg.slogdet()[1]) */
int vilma_num_matrix_log_det(const double *mats, int64_t n, int P, int closed_form, double *out);
/* vi_sigma_inv (numerics.py:247-254): matrices [M,P,P,N] -> out [M,P,P,N] */
int vilma_num_vi_sigma_inv(const double *matrices, int M, int P, int64_t N, double *out);
/* vi_sigma_log_det (numerics.py:283-290): matrices [M,P,P,N] -> out [M,N] */
int vilma_num_vi_sigma_log_det(const double *matrices, int M, int P, int64_t N, double *out);

#ifdef __cplusplus
}
#endif
#endif
