"""ctypes binding of oracle/numerics_omp.c -- TEST INFRASTRUCTURE (see oracle/__init__.py).

`build()` compiles the C / OpenMP restatement of the reference's jitted loops with gcc;
`enable()` swaps the per-SNP functions of oracle.numerics for the compiled ones (same names,
same arguments, same array layouts), `disable()` swaps them back.  bench.py's cpu_baseline leg
times the oracle with them enabled: LD products through BLAS in the reference's per-block loop,
per-SNP passes as compiled loops threaded over all cores -- what numba's `prange` gives the
reference.  Never used by anything under vilma_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import numerics as nm

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'numerics_omp.c')
LIB = os.path.join(HERE, 'libnumerics_omp.so')

_lib = None
_saved = {}
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call(['gcc', '-O3', '-fopenmp', '-fPIC', '-shared',
                           '-o', LIB, SRC, '-lm'])
    return LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB)
        lib.fast_likelihood.restype = C.c_double
        lib.fast_inner_product_comp.restype = C.c_double
        lib.fast_delta_kl.restype = C.c_double
        lib.fast_beta_kl.restype = C.c_double
        _lib = lib
    return _lib


def set_threads(n):
    """OpenMP threads of the compiled loops (libgomp of this process)."""
    C.CDLL('libgomp.so.1').omp_set_num_threads(int(n))


# ---- wrappers: the signatures of oracle.numerics / the reference's numerics.py ----------------
def sum_betas(old_beta, new_beta, step_size):
    o, n = _f(old_beta), _f(new_beta)
    out = np.empty_like(o)
    load().sum_betas(_p(o), _p(n), C.c_double(step_size), C.c_int64(o.size), _p(out))
    return out


def fast_divide(x, y):
    x, y = _f(x), _f(y)
    out = np.empty_like(x)
    load().fast_divide(_p(x), _p(y), C.c_int64(x.size), _p(out))
    return out


def fast_linked_ests(w, x, y, z):
    w, x, y, z = _f(w), _f(x), _f(y), _f(z)
    out = np.empty_like(w)
    load().fast_linked_ests(_p(w), _p(x), _p(y), _p(z), C.c_int64(w.size), _p(out))
    return out


def fast_likelihood(post_means, post_vars, scaled_mu, scaled_ld_diags, linked_ests,
                    adj_marginal, chi_stat, ld_ranks, error_scaling):
    arrs = [_f(a) for a in (post_means, post_vars, scaled_mu, scaled_ld_diags, linked_ests,
                            adj_marginal, chi_stat, ld_ranks, error_scaling)]
    P, N = arrs[0].shape
    return float(load().fast_likelihood(*[_p(a) for a in arrs], C.c_int(P), C.c_int64(N)))


def fast_posterior_mean(vi_mu, vi_delta):
    mu, d = _f(vi_mu), _f(vi_delta)
    M, P, N = mu.shape
    out = np.empty((P, N))
    load().fast_posterior_mean(_p(mu), _p(d), C.c_int(M), C.c_int(P), C.c_int64(N), _p(out))
    return out


def fast_pmv(mean, vi_mu, vi_delta, temp):
    mean, mu, d, temp = _f(mean), _f(vi_mu), _f(vi_delta), _f(temp)
    M, P, N = mu.shape
    out = np.empty((P, N))
    load().fast_pmv(_p(mean), _p(mu), _p(d), _p(temp), C.c_int(M), C.c_int(P), C.c_int64(N),
                    _p(out))
    return out


def _nat(fn, vi_mu, nat_sigma):
    mu, ns = _f(vi_mu), _f(nat_sigma)
    M, P, N = mu.shape
    out = np.empty_like(mu)
    fn(_p(mu), _p(ns), C.c_int(M), C.c_int(P), C.c_int64(N), _p(out))
    return out


def fast_nat_inner_product_m2(vi_mu, nat_sigma):
    return _nat(load().fast_nat_inner_product_m2, vi_mu, nat_sigma)


def fast_nat_inner_product(vi_mu, nat_sigma):
    return _nat(load().fast_nat_inner_product, vi_mu, nat_sigma)


def fast_inner_product_comp(vi_mu, mixture_prec, vi_delta):
    if mixture_prec.shape[-1] != 1:
        raise ValueError('mixture_prec must be 1 dimensional along last mode.')
    mu, pr, d = _f(vi_mu), _f(mixture_prec[:, :, :, 0]), _f(vi_delta)
    M, P, N = mu.shape
    return float(load().fast_inner_product_comp(_p(mu), _p(pr), _p(d), C.c_int(M), C.c_int(P),
                                                C.c_int64(N)))


def sum_annotations(deltas, annotations, num_annotations):
    d = _f(deltas)
    ann = np.ascontiguousarray(annotations, dtype=np.int64)
    out = np.zeros((num_annotations, d.shape[1]))
    load().sum_annotations(_p(d), _i(ann), C.c_int(num_annotations), C.c_int(d.shape[1]),
                           C.c_int64(d.shape[0]), _p(out))
    return out


def fast_delta_kl(vi_delta, hyper_delta, annotations):
    d, h = _f(vi_delta), _f(hyper_delta)
    ann = np.ascontiguousarray(annotations, dtype=np.int64)
    return float(load().fast_delta_kl(_p(d), _p(h), _i(ann), C.c_int(h.shape[0]),
                                      C.c_int(h.shape[1]), C.c_int64(d.shape[0])))


def fast_beta_kl(sigma_summary, vi_delta):
    s, d = _f(sigma_summary), _f(vi_delta)
    return float(load().fast_beta_kl(_p(s), _p(d), C.c_int64(s.size)))


def fast_vi_delta_grad(hyper_delta, log_det, annotations):
    h, ld = _f(hyper_delta), _f(log_det)
    ann = np.ascontiguousarray(annotations, dtype=np.int64)
    out = np.empty((ann.shape[0], h.shape[1] - 1))
    load().fast_vi_delta_grad(_p(h), _p(ld), _i(ann), C.c_int(h.shape[0]), C.c_int(h.shape[1]),
                              C.c_int64(ann.shape[0]), _p(out))
    return out


def fast_invert_nat_vi_delta(new_mu, nat_mu, const_part, nat_vi_delta):
    mu, nat, cp, nv = _f(new_mu), _f(nat_mu), _f(const_part), _f(nat_vi_delta)
    M, P, N = mu.shape
    out = np.empty((N, M))
    load().fast_invert_nat_vi_delta(_p(mu), _p(nat), _p(cp), _p(nv), C.c_int(M), C.c_int(P),
                                    C.c_int64(N), _p(out))
    return out


NATIVE = ('sum_betas', 'fast_divide', 'fast_linked_ests', 'fast_likelihood',
          'fast_posterior_mean', 'fast_pmv', 'fast_nat_inner_product_m2',
          'fast_nat_inner_product', 'fast_inner_product_comp', 'sum_annotations',
          'fast_delta_kl', 'fast_beta_kl', 'fast_vi_delta_grad', 'fast_invert_nat_vi_delta')


def enable(threads=None):
    """Route oracle.numerics' per-SNP functions through the compiled loops."""
    load()
    if threads is not None:
        set_threads(threads)
    for name in NATIVE:
        if name not in _saved:
            _saved[name] = getattr(nm, name)
        setattr(nm, name, globals()[name])


def disable():
    for name, fn in _saved.items():
        setattr(nm, name, fn)
    _saved.clear()
