"""Function API on the GPU (vilma_amd.numerics -> include/vilma_numerics.h) against the
known-answer vectors the reference's own numerics.py produced (tests/golden/numerics_kat.npz),
against the oracle on seeded inputs at sizes that straddle the kernels' tiles, and through the
recipes of the reference's tests (/root/reference/tests/test.py:877-1217)."""
import numpy as np
import pytest

from helpers import golden
from oracle import numerics as onm
from vilma_amd import numerics as nm

pytestmark = pytest.mark.gpu

K = golden('numerics_kat.npz')
RTOL, ATOL = 1e-11, 1e-13


@pytest.mark.parametrize('P', [1, 2, 3])
def test_reference_known_answers(P):
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
        np.testing.assert_allclose(got, g(name), rtol=RTOL, atol=ATOL, err_msg=name)


def _spd_stack(rng, lead, P):
    x = rng.random(lead + (P, P))
    x = x + np.swapaxes(x, -1, -2)
    x[..., np.arange(P), np.arange(P)] += 3
    return x


# (M, P, N, A): tile edges (KT = 16 components, 128 SNPs per workgroup), one component, ragged N
SHAPES = [(1, 1, 1, 1), (2, 1, 127, 1), (16, 2, 128, 2), (17, 3, 129, 3), (33, 4, 1000, 2),
          (40, 2, 5003, 1), (81, 4, 2050, 4)]


@pytest.mark.parametrize('M,P,N,A', SHAPES)
def test_against_the_oracle(M, P, N, A):
    rng = np.random.default_rng(M * 1000 + P * 100 + A)
    mu, mu2 = rng.normal(size=(M, P, N)), rng.normal(size=(M, P, N))
    delta = rng.random((N, M)) + 1e-3
    delta /= delta.sum(axis=1, keepdims=True)
    ann = rng.integers(0, A, N).astype(np.int64)
    hyper = rng.random((A, M)) + 0.05
    hyper /= hyper.sum(axis=1, keepdims=True)
    x, y, w, z = (rng.random((P, N)) + 0.1 for _ in range(4))
    lam = np.ascontiguousarray(np.transpose(_spd_stack(rng, (M, N), P), (0, 2, 3, 1)))
    prec = _spd_stack(rng, (M,), P)[..., None]
    const, natd = rng.normal(size=(N, M)), rng.normal(size=(N, M - 1))
    log_det, chi, ranks, tau = rng.normal(size=M), rng.normal(size=P), rng.random(P) * N, \
        rng.random(P) + 0.5
    pairs = [
        ('sum_betas', (mu, mu2, 0.37)),
        ('fast_divide', (x, y)),
        ('fast_linked_ests', (w, y, x, z)),
        ('fast_likelihood', (x, y, w, z, mu[0], mu2[0], chi, ranks, tau)),
        ('fast_posterior_mean', (mu, delta)),
        ('fast_pmv', (onm.fast_posterior_mean(mu, delta), mu, delta, np.abs(mu2))),
        ('fast_nat_inner_product_m2', (mu, lam)),
        ('fast_nat_inner_product', (mu, lam)),
        ('fast_inner_product_comp', (mu, prec, delta)),
        ('sum_annotations', (delta, ann, A)),
        ('fast_delta_kl', (delta, hyper, ann)),
        ('fast_beta_kl', (const, delta)),
        ('fast_vi_delta_grad', (hyper, log_det, ann)),
        ('map_to_nat_cat_2D', (delta,)),
        ('invert_nat_cat_2D', (natd * 30,)),
        ('fast_invert_nat_vi_delta', (mu, mu2, const, natd)),
        ('vi_sigma_inv', (lam,)),
        ('vi_sigma_log_det', (lam,)),
    ]
    for name, args in pairs:
        want = getattr(onm, name)(*args)
        got = getattr(nm, name)(*args)
        assert np.shape(got) == np.shape(want), name
        scale = max(1.0, float(np.max(np.abs(want)))) if np.size(want) else 1.0
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 * scale, err_msg=name)


def test_results_do_not_depend_on_scheduling():
    """Two-stage reductions with a fixed order: the same call returns the same bits."""
    rng = np.random.default_rng(5)
    mu = rng.normal(size=(40, 2, 30011))
    delta = rng.random((30011, 40))
    prec = _spd_stack(rng, (40,), 2)[..., None]
    ann = rng.integers(0, 3, 30011).astype(np.int64)
    a = [nm.fast_inner_product_comp(mu, prec, delta), nm.fast_beta_kl(delta, delta)]
    b = [nm.fast_inner_product_comp(mu, prec, delta), nm.fast_beta_kl(delta, delta)]
    assert a == b
    s1, s2 = nm.sum_annotations(delta, ann, 3), nm.sum_annotations(delta, ann, 3)
    assert np.array_equal(s1, s2)
    np.testing.assert_allclose(s1.sum(axis=0), delta.sum(axis=0), rtol=1e-12)


def test_invert_nat_cat_clamps_without_renormalising():
    # numerics.py:184-194: entries below 1e-100 are clamped, the row is not renormalised
    out = nm.invert_nat_cat_2D(np.array([[800.0, -800.0, 0.0]]))
    assert out[0, 0] == 1.0 and out[0, 1] == 1e-100 and out[0, 3] == 1e-100
    assert out[0, 2] == 1e-100


def test_nat_cat_round_trip():
    # the recipe of the reference's test_invert_nat_cat_2D (tests/test.py:1068-1081)
    rng = np.random.default_rng(11)
    x = rng.random((5, 10))
    x /= x.sum(axis=1, keepdims=True)
    np.testing.assert_allclose(nm.invert_nat_cat_2D(nm.map_to_nat_cat_2D(x)), x, rtol=1e-12)
    nat = rng.normal(size=(300, 37))
    ext = np.concatenate([nat, np.zeros((300, 1))], axis=1)
    true = np.exp(ext)
    true /= true.sum(axis=1, keepdims=True)
    np.testing.assert_allclose(nm.invert_nat_cat_2D(nat), true, rtol=1e-12)


def test_fast_inner_product_comp_rejects_per_snp_precisions():
    # tests/test.py:1011-1013
    rng = np.random.default_rng(2)
    with pytest.raises(ValueError):
        nm.fast_inner_product_comp(rng.normal(size=(3, 2, 100)), rng.random((3, 2, 2, 100)),
                                   rng.random((100, 3)))


@pytest.mark.parametrize('P', [1, 2, 3, 4, 5, 8])
def test_matrix_invert_and_log_det_against_lapack(P):
    # the recipes of tests/test.py:1107-1203: every leading shape the reference test uses
    rng = np.random.default_rng(P)
    for lead in [(), (10,), (2, 10), (2, 2, 10)]:
        x = _spd_stack(rng, lead, P)
        np.testing.assert_allclose(nm.matrix_invert(x), np.linalg.inv(x), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(nm.matrix_log_det(x), np.linalg.slogdet(x)[1], rtol=1e-11,
                                   atol=1e-13)
    # a general (non-symmetric, pivoting) matrix outside the 4-D special case
    g = rng.normal(size=(50, P, P))
    g[:, 0, 0] = 1e-9                      # forces a row exchange for P > 1
    np.testing.assert_allclose(nm.matrix_invert(g) @ g, np.broadcast_to(np.eye(P), g.shape),
                               rtol=0, atol=1e-6 if P > 1 else 1e-12)
    np.testing.assert_allclose(nm.matrix_log_det(g), np.linalg.slogdet(g)[1], rtol=1e-9,
                               atol=1e-9)
    lam = np.ascontiguousarray(np.transpose(_spd_stack(rng, (3, 100), P), (0, 2, 3, 1)))
    true_inv = np.array([np.linalg.inv(m.T).T for m in lam])
    np.testing.assert_allclose(nm.vi_sigma_inv(lam), true_inv, rtol=1e-11, atol=1e-13)
    true_ld = np.array([np.linalg.slogdet(m.T)[1].T for m in lam])
    np.testing.assert_allclose(nm.vi_sigma_log_det(lam), true_ld, rtol=1e-11, atol=1e-13)


def test_two_by_two_special_case_is_written_symmetric():
    # numerics.py:231-232: the 4-D 2 x 2 path sets inv[1,0] = inv[0,1]
    x = np.array([[[[2.0, 1.0], [0.5, 3.0]]]])
    got = nm._matrix_invert_4d_numba(x)
    det = 1. / (2.0 * 3.0 - 1.0 * 0.5)
    assert np.allclose(got[0, 0], [[3.0 * det, -1.0 * det], [-1.0 * det, 2.0 * det]])
    with pytest.raises(ValueError):
        nm._matrix_invert_4d_numba(np.ones((2, 2, 3, 3)))
    with pytest.raises(ValueError):
        nm._matrix_log_det_4d_numba(np.ones((2, 2, 3, 3)))


def test_likelihood_recipe_of_the_reference_test():
    # tests/test.py:901-944 with diagonal LD
    rng = np.random.default_rng(3)
    mu = rng.normal(size=(3, 2, 100))
    diags = rng.random((3, 2, 100))
    std_errs = rng.random((2, 100)) + 0.1
    delta = rng.random((100, 3))
    delta /= delta.sum(axis=1, keepdims=True)
    tau, ranks, chi = rng.random(2) + 0.1, np.array([100., 100.]), rng.normal(size=2)
    adj, ld_diags = rng.normal(size=(2, 100)), rng.random((2, 100))
    pm = nm.fast_posterior_mean(mu, delta)
    pv = nm.fast_pmv(pm, mu, delta, diags)
    smu = nm.fast_divide(pm, std_errs)
    linked = ld_diags * smu
    true = 0.
    for p in range(2):
        t = -0.5 * (ld_diags[p] * pv[p] * std_errs[p] ** -2).sum()
        t += -0.5 * smu[p].dot(ld_diags[p] * smu[p]) + pm[p].dot(adj[p])
        true += t / tau[p] - 0.5 * ranks[p] * np.log(tau[p]) - 0.5 * chi[p] / tau[p]
    got = nm.fast_likelihood(pm, pv, smu, ld_diags / std_errs ** 2, linked, adj, chi, ranks, tau)
    assert np.isclose(true, got, rtol=1e-12)
