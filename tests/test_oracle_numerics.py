"""Oracle numerics vs known-answer vectors produced by the reference's numerics.py
(tests/golden/make_golden.py::numerics_kat; recipes follow reference tests/test.py:877-1217)."""
import numpy as np
import pytest

from helpers import golden
from oracle import numerics as nm

K = golden('numerics_kat.npz')


@pytest.mark.parametrize('P', [1, 2, 3])
def test_numerics_kat(P):
    t = 'P%d_' % P
    g = lambda n: K[t + n]
    A = g('hyper').shape[0]
    checks = {
        'sum_betas': nm.sum_betas(g('mu'), g('mu2'), 0.3),
        'fast_divide': nm.fast_divide(g('x'), g('y')),
        'fast_linked_ests': nm.fast_linked_ests(g('w'), g('y'), g('x'), g('z')),
        'fast_likelihood': nm.fast_likelihood(g('x'), g('y'), g('w'), g('z'), g('mu')[0],
                                              g('mu2')[0], g('chi'), g('ranks'), g('tau')),
        'fast_posterior_mean': nm.fast_posterior_mean(g('mu'), g('delta')),
        'fast_pmv': nm.fast_pmv(nm.fast_posterior_mean(g('mu'), g('delta')), g('mu'),
                                g('delta'), np.abs(g('mu2'))),
        'fast_nat_inner_product_m2': nm.fast_nat_inner_product_m2(g('mu'), g('lam')),
        'fast_nat_inner_product': nm.fast_nat_inner_product(g('mu'), g('lam')),
        'fast_inner_product_comp': nm.fast_inner_product_comp(g('mu'), g('prec'), g('delta')),
        'sum_annotations': nm.sum_annotations(g('delta'), g('ann'), A),
        'fast_delta_kl': nm.fast_delta_kl(g('delta'), g('hyper'), g('ann')),
        'fast_beta_kl': nm.fast_beta_kl(g('const'), g('delta')),
        'fast_vi_delta_grad': nm.fast_vi_delta_grad(g('hyper'), g('log_det'), g('ann')),
        'map_to_nat_cat_2D': nm.map_to_nat_cat_2D(g('delta')),
        'invert_nat_cat_2D': nm.invert_nat_cat_2D(g('natd') * 40),
        'fast_invert_nat_vi_delta': nm.fast_invert_nat_vi_delta(g('mu'), g('mu2'), g('const'),
                                                               g('natd')),
        'vi_sigma_inv': nm.vi_sigma_inv(g('lam')),
        'vi_sigma_log_det': nm.vi_sigma_log_det(g('lam')),
    }
    for name, got in checks.items():
        np.testing.assert_allclose(got, g(name), rtol=1e-11, atol=1e-13, err_msg=name)


def test_invert_nat_cat_clamps_without_renormalising():
    # numerics.py:184-194: entries below 1e-100 are clamped, the row is not renormalised
    logits = np.array([[800.0, -800.0, 0.0]])
    out = nm.invert_nat_cat_2D(logits)
    assert out[0, 0] == 1.0 and out[0, 1] == 1e-100 and out[0, 3] == 1e-100
    assert out[0, 2] == 1e-100        # exp(-800) underflows to 0 and is clamped too
