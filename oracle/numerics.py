"""Oracle restatement of the reference's jitted inner functions (numerics.py).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Each function follows the reference
function of the same name in /root/reference/src/vilma/numerics.py (line ranges cited);
the per-SNP `prange` loops are written as whole-array numpy expressions.  Layouts are the
reference's: vi_mu [M,P,N], vi_delta [N,M], vi_sigma/nat_sigma [M,P,P,N], hyper_delta [A,M].
"""
import numpy as np

EPSILON = 1e-100    # numerics.py:8


def sum_betas(old_beta, new_beta, step_size):
    """numerics.py:11-15 -- convex blend of natural parameters."""
    return step_size * new_beta + (1. - step_size) * old_beta


def fast_divide(x, y):
    """numerics.py:18-21."""
    return x / y


def fast_linked_ests(w, x, y, z):
    """numerics.py:24-28 -- w/x - y*z."""
    return w / x - y * z


def fast_likelihood(post_means, post_vars, scaled_mu, scaled_ld_diags, linked_ests,
                    adj_marginal, chi_stat, ld_ranks, error_scaling):
    """numerics.py:31-46 -- expected log likelihood, summed over cohorts."""
    per_pop = (-0.5 * (scaled_ld_diags * post_vars + linked_ests * scaled_mu)
               + post_means * adj_marginal).sum(axis=1)
    per_pop = per_pop - 0.5 * chi_stat
    return float((per_pop / error_scaling
                  - 0.5 * ld_ranks * np.log(error_scaling)).sum())


def fast_posterior_mean(vi_mu, vi_delta):
    """numerics.py:49-57 -- m[p,i] = sum_k mu[k,p,i] delta[i,k]."""
    return np.einsum('kpi,ik->pi', vi_mu, vi_delta)


def fast_pmv(mean, vi_mu, vi_delta, temp):
    """numerics.py:60-65 -- posterior marginal variance."""
    return fast_posterior_mean(temp + vi_mu ** 2, vi_delta) - mean ** 2


def fast_nat_inner_product_m2(vi_mu, nat_sigma):
    """numerics.py:68-80 -- -2 * 'sqi,spqi->spi'."""
    return -2. * np.einsum('spqi,sqi->spi', nat_sigma, vi_mu)


def fast_nat_inner_product(vi_mu, nat_sigma):
    """numerics.py:83-95 -- 'sqi,spqi->spi'."""
    return np.einsum('spqi,sqi->spi', nat_sigma, vi_mu)


def fast_inner_product_comp(vi_mu, mixture_prec, vi_delta):
    """numerics.py:98-115 -- 0.5 * sum_ik delta_ik mu_ki^T Prec_k mu_ki."""
    if mixture_prec.shape[-1] != 1:
        raise ValueError('mixture_prec must be 1 dimensional along last mode.')
    quad = np.einsum('kpi,kqi,kqp->ik', vi_mu, vi_mu, mixture_prec[:, :, :, 0])
    return 0.5 * float((quad * vi_delta).sum())


def sum_annotations(deltas, annotations, num_annotations):
    """numerics.py:118-129 -- per-annotation column sums of vi_delta."""
    out = np.zeros((num_annotations, deltas.shape[1]))
    for a in range(num_annotations):
        out[a] = deltas[annotations == a].sum(axis=0)
    return out


def fast_delta_kl(vi_delta, hyper_delta, annotations):
    """numerics.py:132-141 -- sum_ik delta (log delta - log hyper[a_i])."""
    log_hyper = np.log(hyper_delta)
    return float((vi_delta * (np.log(vi_delta) - log_hyper[annotations])).sum())


def fast_beta_kl(sigma_summary, vi_delta):
    """numerics.py:144-146."""
    return 0.5 * float((sigma_summary * vi_delta).sum())


def fast_vi_delta_grad(hyper_delta, log_det, annotations):
    """numerics.py:149-164 -- natural parameter of the mixture weights, last entry pivot."""
    full = np.log(hyper_delta) - 0.5 * log_det[None, :]
    per_snp = full[annotations]
    return per_snp[:, :-1] - per_snp[:, -1:]


def map_to_nat_cat_2D(probs):
    """numerics.py:167-176."""
    logp = np.log(probs)
    return logp[:, :-1] - logp[:, -1:]


def invert_nat_cat_2D(probs):
    """numerics.py:179-195 -- softmax with implicit last logit 0, max-trick with
    max(max logit, 0); each entry clamped at EPSILON and NOT renormalised."""
    max_p = np.maximum(probs.max(axis=1), 0.)[:, None] if probs.shape[1] else \
        np.zeros((probs.shape[0], 1))
    last = np.exp(-max_p)
    this = np.exp(probs - max_p)
    denom = last + this.sum(axis=1, keepdims=True)
    out = np.empty((probs.shape[0], probs.shape[1] + 1))
    out[:, :-1] = np.maximum(this / denom, EPSILON)
    out[:, -1:] = np.maximum(last / denom, EPSILON)
    return out


def fast_invert_nat_vi_delta(new_mu, nat_mu, const_part, nat_vi_delta):
    """numerics.py:198-213 -- logits from (mu, nat_mu) then invert_nat_cat_2D."""
    addenda = const_part + np.einsum('kji,kji->ik', new_mu, nat_mu)
    logits = 0.5 * (addenda[:, :-1] - addenda[:, -1:]) + nat_vi_delta
    return invert_nat_cat_2D(logits)


def vi_sigma_inv(matrices):
    """numerics.py:216-254 -- invert the middle PxP of a [M,P,P,N] array; closed form for
    P<=2 (as the reference's numba helper), LAPACK otherwise."""
    P = matrices.shape[1]
    if P == 1:
        return 1. / matrices
    if P == 2:
        a, b = matrices[:, 0, 0], matrices[:, 0, 1]
        c, d = matrices[:, 1, 0], matrices[:, 1, 1]
        det = 1. / (a * d - b * c)
        out = np.empty_like(matrices)
        out[:, 0, 0] = d * det
        out[:, 1, 1] = a * det
        # the reference mirrors the (0,1) entry of the *transposed* view (numerics.py:230-231)
        out[:, 1, 0] = -c * det
        out[:, 0, 1] = out[:, 1, 0]
        return out
    return np.transpose(np.linalg.inv(np.transpose(matrices, (3, 0, 1, 2))), (1, 2, 3, 0))


def vi_sigma_log_det(matrices):
    """numerics.py:257-290 -- log-determinants [M,N] of a [M,P,P,N] array."""
    P = matrices.shape[1]
    if P == 1:
        return np.log(matrices[:, 0, 0])
    if P == 2:
        return np.log(matrices[:, 0, 0] * matrices[:, 1, 1]
                      - matrices[:, 0, 1] * matrices[:, 1, 0])
    return np.transpose(np.linalg.slogdet(np.transpose(matrices, (3, 0, 1, 2)))[1])
