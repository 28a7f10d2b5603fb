"""The reference's Function API (`vilma.numerics`) on the GPU.

Same names, argument order, layouts and return shapes as
/root/reference/src/vilma/numerics.py:11-290; each function is one synchronous call into
libvilma_hip.so (include/vilma_numerics.h, csrc/numerics_api.hip) that copies the arguments to
the device, runs hand-written kernels and returns a fresh array.  There is no host arithmetic
here: without the library or a GPU every call raises.

The fit does not go through these -- `MultiPopVI` keeps its state on the device and evaluates a
whole point in fused kernels -- so moving [M,P,N] arrays per call is the
price of calling the decomposed functions directly, exactly as their signatures demand.

Like the reference's numba signatures, arguments must be float64 (int64 for annotations) with
the stated number of dimensions; anything else raises TypeError.
"""
import numpy as np

from . import _lib

EPSILON = 1e-100        # numerics.py:8


def _f64(x, ndim, name):
    if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim != ndim:
        raise TypeError('%s must be a %d-dimensional float64 array' % (name, ndim))
    return np.ascontiguousarray(x)


def _i64(x, ndim, name):
    if not isinstance(x, np.ndarray) or x.dtype != np.int64 or x.ndim != ndim:
        raise TypeError('%s must be a %d-dimensional int64 array' % (name, ndim))
    return np.ascontiguousarray(x)


def _same(name, *arrays):
    for a in arrays[1:]:
        if a.shape != arrays[0].shape:
            raise ValueError('%s: argument shapes differ: %s'
                             % (name, ', '.join(str(x.shape) for x in arrays)))


def _call(fn_name, *args):
    lib = _lib.load()
    conv = [a.ctypes.data if isinstance(a, np.ndarray) else a for a in args]
    if getattr(lib, fn_name)(*conv) != 0:
        raise _lib.VilmaHipError('%s: %s' % (fn_name, lib.vilma_num_last_error().decode()))


def sum_betas(old_beta, new_beta, step_size):
    """numerics.py:11-15 -- step_size * new_beta + (1 - step_size) * old_beta, [M,P,N]."""
    old_beta, new_beta = _f64(old_beta, 3, 'old_beta'), _f64(new_beta, 3, 'new_beta')
    _same('sum_betas', old_beta, new_beta)
    out = np.empty_like(old_beta)
    _call('vilma_num_sum_betas', old_beta, new_beta, float(step_size), old_beta.size, out)
    return out


def fast_divide(x, y):
    """numerics.py:18-21 -- x / y, [P,N]."""
    x, y = _f64(x, 2, 'x'), _f64(y, 2, 'y')
    _same('fast_divide', x, y)
    out = np.empty_like(x)
    _call('vilma_num_divide', x, y, x.size, out)
    return out


def fast_linked_ests(w, x, y, z):
    """numerics.py:24-28 -- w / x - y * z, [P,N]."""
    w, x, y, z = (_f64(a, 2, n) for a, n in zip((w, x, y, z), 'wxyz'))
    _same('fast_linked_ests', w, x, y, z)
    out = np.empty_like(w)
    _call('vilma_num_linked_ests', w, x, y, z, w.size, out)
    return out


def fast_likelihood(post_means, post_vars, scaled_mu, scaled_ld_diags, linked_ests,
                    adj_marginal, chi_stat, ld_ranks, error_scaling):
    """numerics.py:31-46 -- expected log likelihood summed over cohorts."""
    names = ('post_means', 'post_vars', 'scaled_mu', 'scaled_ld_diags', 'linked_ests',
             'adj_marginal')
    mats = [_f64(a, 2, n) for a, n in zip((post_means, post_vars, scaled_mu, scaled_ld_diags,
                                           linked_ests, adj_marginal), names)]
    _same('fast_likelihood', *mats)
    vecs = [_f64(a, 1, n) for a, n in zip((chi_stat, ld_ranks, error_scaling),
                                          ('chi_stat', 'ld_ranks', 'error_scaling'))]
    P, N = mats[0].shape
    for v in vecs:
        if v.shape != (P,):
            raise ValueError('fast_likelihood: per-cohort vectors must have length %d' % P)
    out = np.zeros(1)
    _call('vilma_num_likelihood', *mats, *vecs, P, N, out)
    return float(out[0])


def _mu_delta(vi_mu, vi_delta, name):
    vi_mu, vi_delta = _f64(vi_mu, 3, 'vi_mu'), _f64(vi_delta, 2, 'vi_delta')
    M, P, N = vi_mu.shape
    if vi_delta.shape != (N, M):
        raise ValueError('%s: vi_delta must be [N,M] = (%d, %d), not %s'
                         % (name, N, M, vi_delta.shape))
    return vi_mu, vi_delta, M, P, N


def fast_posterior_mean(vi_mu, vi_delta):
    """numerics.py:49-57 -- m[p,i] = sum_k vi_mu[k,p,i] vi_delta[i,k]."""
    vi_mu, vi_delta, M, P, N = _mu_delta(vi_mu, vi_delta, 'fast_posterior_mean')
    out = np.zeros((P, N))
    _call('vilma_num_posterior_mean', vi_mu, vi_delta, M, P, N, out)
    return out


def fast_pmv(mean, vi_mu, vi_delta, temp):
    """numerics.py:60-65 -- posterior marginal variance from the mean and diag(vi_sigma)."""
    vi_mu, vi_delta, M, P, N = _mu_delta(vi_mu, vi_delta, 'fast_pmv')
    mean, temp = _f64(mean, 2, 'mean'), _f64(temp, 3, 'temp')
    if mean.shape != (P, N) or temp.shape != (M, P, N):
        raise ValueError('fast_pmv: mean must be [P,N] and temp [M,P,N]')
    out = np.zeros((P, N))
    _call('vilma_num_pmv', mean, vi_mu, vi_delta, temp, M, P, N, out)
    return out


def _nat_inner(vi_mu, nat_sigma, scale, name):
    vi_mu, nat_sigma = _f64(vi_mu, 3, 'vi_mu'), _f64(nat_sigma, 4, 'nat_sigma')
    M, P, N = vi_mu.shape
    if nat_sigma.shape != (M, P, P, N):
        raise ValueError('%s: nat_sigma must be [M,P,P,N]' % name)
    out = np.zeros((M, P, N))
    _call('vilma_num_nat_inner_product', vi_mu, nat_sigma, M, P, N, scale, out)
    return out


def fast_nat_inner_product_m2(vi_mu, nat_sigma):
    """numerics.py:68-80 -- -2 * einsum('sqi,spqi->spi', vi_mu, nat_sigma)."""
    return _nat_inner(vi_mu, nat_sigma, -2.0, 'fast_nat_inner_product_m2')


def fast_nat_inner_product(vi_mu, nat_sigma):
    """numerics.py:83-95 -- einsum('sqi,spqi->spi', vi_mu, nat_sigma)."""
    return _nat_inner(vi_mu, nat_sigma, 1.0, 'fast_nat_inner_product')


def fast_inner_product_comp(vi_mu, mixture_prec, vi_delta):
    """numerics.py:98-115 -- 0.5 * sum_ik vi_delta[i,k] mu_ki^T Prec_k mu_ki; mixture_prec is
    [M,P,P,1] as the reference stores it."""
    vi_mu, vi_delta, M, P, N = _mu_delta(vi_mu, vi_delta, 'fast_inner_product_comp')
    mixture_prec = _f64(mixture_prec, 4, 'mixture_prec')
    if mixture_prec.shape[-1] != 1:
        raise ValueError('mixture_prec must be 1 dimensional along last mode.')
    if mixture_prec.shape != (M, P, P, 1):
        raise ValueError('fast_inner_product_comp: mixture_prec must be [M,P,P,1]')
    out = np.zeros(1)
    _call('vilma_num_inner_product_comp', vi_mu, mixture_prec, vi_delta, M, P, N, out)
    return float(out[0])


def sum_annotations(deltas, annotations, num_annotations):
    """numerics.py:118-129 -- per-annotation column sums of vi_delta, [A,M]."""
    deltas, annotations = _f64(deltas, 2, 'deltas'), _i64(annotations, 1, 'annotations')
    N, M = deltas.shape
    if annotations.shape != (N,):
        raise ValueError('sum_annotations: annotations must have one entry per row of deltas')
    A = int(num_annotations)
    out = np.zeros((A, M))
    _call('vilma_num_sum_annotations', deltas, annotations, A, M, N, out)
    return out


def fast_delta_kl(vi_delta, hyper_delta, annotations):
    """numerics.py:132-141 -- sum_ik delta_ik (log delta_ik - log hyper[a_i,k])."""
    vi_delta, hyper_delta = _f64(vi_delta, 2, 'vi_delta'), _f64(hyper_delta, 2, 'hyper_delta')
    annotations = _i64(annotations, 1, 'annotations')
    N, M = vi_delta.shape
    if hyper_delta.shape[1] != M or annotations.shape != (N,):
        raise ValueError('fast_delta_kl: hyper_delta must be [A,M] and annotations [N]')
    out = np.zeros(1)
    _call('vilma_num_delta_kl', vi_delta, hyper_delta, annotations, hyper_delta.shape[0], M, N,
          out)
    return float(out[0])


def fast_beta_kl(sigma_summary, vi_delta):
    """numerics.py:144-146 -- 0.5 * sum(sigma_summary * vi_delta)."""
    sigma_summary = _f64(sigma_summary, 2, 'sigma_summary')
    vi_delta = _f64(vi_delta, 2, 'vi_delta')
    _same('fast_beta_kl', sigma_summary, vi_delta)
    out = np.zeros(1)
    _call('vilma_num_beta_kl', sigma_summary, vi_delta, vi_delta.size, out)
    return float(out[0])


def fast_vi_delta_grad(hyper_delta, log_det, annotations):
    """numerics.py:149-164 -- natural gradient of vi_delta, [N, M-1]."""
    hyper_delta, log_det = _f64(hyper_delta, 2, 'hyper_delta'), _f64(log_det, 1, 'log_det')
    annotations = _i64(annotations, 1, 'annotations')
    A, M = hyper_delta.shape
    if log_det.shape != (M,):
        raise ValueError('fast_vi_delta_grad: log_det must have one entry per component')
    N = annotations.shape[0]
    out = np.zeros((N, M - 1))
    _call('vilma_num_vi_delta_grad', hyper_delta, log_det, annotations, A, M, N, out)
    return out


def map_to_nat_cat_2D(probs):
    """numerics.py:167-176 -- log(probs[i,k] / probs[i,-1]) for all but the last column."""
    probs = _f64(probs, 2, 'probs')
    N, K = probs.shape
    out = np.zeros((N, K - 1))
    _call('vilma_num_map_to_nat_cat', probs, N, K, out)
    return out


def invert_nat_cat_2D(probs):
    """numerics.py:179-195 -- back to probabilities (max trick, clamped at 1e-100, not
    renormalised)."""
    probs = _f64(probs, 2, 'probs')
    N, K = probs.shape
    out = np.empty((N, K + 1))
    _call('vilma_num_invert_nat_cat', probs, N, K, out)
    return out


def fast_invert_nat_vi_delta(new_mu, nat_mu, const_part, nat_vi_delta):
    """numerics.py:198-213 -- vi_delta [N,M] from its natural parameters."""
    new_mu, nat_mu = _f64(new_mu, 3, 'new_mu'), _f64(nat_mu, 3, 'nat_mu')
    const_part = _f64(const_part, 2, 'const_part')
    nat_vi_delta = _f64(nat_vi_delta, 2, 'nat_vi_delta')
    _same('fast_invert_nat_vi_delta', new_mu, nat_mu)
    M, P, N = new_mu.shape
    if const_part.shape != (N, M) or nat_vi_delta.shape != (N, M - 1):
        raise ValueError('fast_invert_nat_vi_delta: const_part must be [N,M] and nat_vi_delta '
                         '[N,M-1]')
    out = np.empty((N, M))
    _call('vilma_num_invert_nat_vi_delta', new_mu, nat_mu, const_part, nat_vi_delta, M, P, N, out)
    return out


def _check_p(P, what):
    if P > 8:
        raise NotImplementedError('%s: matrices larger than 8 x 8 are not supported by the HIP '
                                  'kernels (the fit is limited to 8 cohorts)' % what)


def _stack_of_matrices(matrix, what):
    if not isinstance(matrix, np.ndarray) or matrix.dtype != np.float64 or matrix.ndim < 2 \
            or matrix.shape[-1] != matrix.shape[-2]:
        raise TypeError('%s needs a float64 array of square matrices [..., P, P]' % what)
    return np.ascontiguousarray(matrix)


def _matrix_invert_4d_numba(matrix):
    """numerics.py:216-235 -- closed-form inverse of [.,.,P,P] stacks, P <= 2."""
    matrix = _f64(matrix, 4, 'matrix')
    P = matrix.shape[-1]
    if P == 0:
        return np.zeros_like(matrix)
    if P > 2:
        raise ValueError('_matrix_invert_4d_numba cannot be used on matrices larger than 2x2')
    out = np.empty_like(matrix)
    _call('vilma_num_matrix_invert', matrix, matrix.size // (P * P), P, 1, out)
    return out


def matrix_invert(matrix):
    """numerics.py:238-244 -- batched inverse over the last two axes."""
    matrix = _stack_of_matrices(matrix, 'matrix_invert')
    P = matrix.shape[-1]
    if P <= 2 and matrix.ndim == 4:
        return _matrix_invert_4d_numba(matrix)
    _check_p(P, 'matrix_invert')
    out = np.empty_like(matrix)
    if P:
        _call('vilma_num_matrix_invert', matrix, matrix.size // (P * P), P, 0, out)
    return out


def vi_sigma_inv(matrices):
    """numerics.py:247-254 -- inverse of every [P,P] slice of an [M,P,P,N] array."""
    matrices = _f64(matrices, 4, 'matrices')
    M, P, P2, N = matrices.shape
    if P != P2:
        raise ValueError('vi_sigma_inv: matrices must be [M,P,P,N]')
    _check_p(P, 'vi_sigma_inv')
    out = np.zeros_like(matrices)
    _call('vilma_num_vi_sigma_inv', matrices, M, P, N, out)
    return out


def _matrix_log_det_4d_numba(matrix):
    """numerics.py:257-271 -- closed-form log determinant of [.,.,P,P] stacks, P <= 2."""
    matrix = _f64(matrix, 4, 'matrix')
    P = matrix.shape[-1]
    if P == 0:
        return np.zeros(matrix.shape[:2])
    if P > 2:
        raise ValueError('_matrix_log_det_4d_numba cannot be used on matrices larger than 2x2')
    out = np.empty(matrix.shape[:2])
    _call('vilma_num_matrix_log_det', matrix, out.size, P, 1, out)
    return out


def matrix_log_det(matrix):
    """numerics.py:274-280 -- batched log|det| over the last two axes."""
    matrix = _stack_of_matrices(matrix, 'matrix_log_det')
    P = matrix.shape[-1]
    if P <= 2 and matrix.ndim == 4:
        return _matrix_log_det_4d_numba(matrix)
    _check_p(P, 'matrix_log_det')
    out = np.zeros(matrix.shape[:-2])
    if P:
        _call('vilma_num_matrix_log_det', matrix, out.size, P, 0, out)
    return out


def vi_sigma_log_det(matrices):
    """numerics.py:283-290 -- log determinants of an [M,P,P,N] array, [M,N]."""
    matrices = _f64(matrices, 4, 'matrices')
    M, P, P2, N = matrices.shape
    if P != P2:
        raise ValueError('vi_sigma_log_det: matrices must be [M,P,P,N]')
    _check_p(P, 'vi_sigma_log_det')
    out = np.zeros((M, N))
    if P:
        _call('vilma_num_vi_sigma_log_det', matrices, M, P, N, out)
    return out
