"""The C / OpenMP restatement of the reference's jitted loops (oracle/numerics_omp.c, the
per-SNP half of bench.py's CPU baseline) against the vectors the reference itself produced
(tests/golden/numerics_kat.npz) and against the numpy restatement, and a whole trajectory
with it enabled."""
import numpy as np
import pytest

from helpers import golden, oracle_from_traj


@pytest.fixture(scope='module')
def native():
    from oracle import native as nat
    nat.build()
    nat.set_threads(4)
    yield nat
    nat.disable()


def test_native_functions_match_numpy_restatement(native):
    from oracle import numerics as nm
    rng = np.random.default_rng(0)
    for M, P, N, A in ((3, 1, 17, 1), (5, 2, 400, 3), (9, 4, 123, 2)):
        mu, mu2 = rng.normal(size=(M, P, N)), rng.normal(size=(M, P, N))
        delta = rng.dirichlet(np.ones(M), size=N)
        sig = rng.normal(size=(M, P, P, N))
        ann = rng.integers(0, A, size=N)
        hyper = rng.dirichlet(np.ones(M), size=A)
        log_det = rng.normal(size=M)
        pn = [rng.normal(size=(P, N)) for _ in range(6)]
        vec = [rng.uniform(0.5, 2, size=P) for _ in range(3)]
        prec = rng.normal(size=(M, P, P, 1))
        pairs = [
            (native.sum_betas(mu, mu2, 0.3), nm.sum_betas(mu, mu2, 0.3)),
            (native.fast_divide(pn[0], pn[1]), nm.fast_divide(pn[0], pn[1])),
            (native.fast_linked_ests(*pn[:4]), nm.fast_linked_ests(*pn[:4])),
            (native.fast_likelihood(*pn, *vec), nm.fast_likelihood(*pn, *vec)),
            (native.fast_posterior_mean(mu, delta), nm.fast_posterior_mean(mu, delta)),
            (native.fast_pmv(pn[0], mu, delta, np.abs(mu2)), nm.fast_pmv(pn[0], mu, delta, np.abs(mu2))),
            (native.fast_nat_inner_product_m2(mu, sig), nm.fast_nat_inner_product_m2(mu, sig)),
            (native.fast_nat_inner_product(mu, sig), nm.fast_nat_inner_product(mu, sig)),
            (native.fast_inner_product_comp(mu, prec, delta), nm.fast_inner_product_comp(mu, prec, delta)),
            (native.sum_annotations(delta, ann, A), nm.sum_annotations(delta, ann, A)),
            (native.fast_delta_kl(delta, hyper, ann), nm.fast_delta_kl(delta, hyper, ann)),
            (native.fast_beta_kl(delta * 3, delta), nm.fast_beta_kl(delta * 3, delta)),
            (native.fast_vi_delta_grad(hyper, log_det, ann), nm.fast_vi_delta_grad(hyper, log_det, ann)),
            (native.fast_invert_nat_vi_delta(mu, mu2, delta * 2, rng.normal(size=(N, M - 1)) * 0 + 0.1),
             nm.fast_invert_nat_vi_delta(mu, mu2, delta * 2, np.full((N, M - 1), 0.1))),
        ]
        for got, want in pairs:
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('P', [1, 2, 3])
def test_native_functions_match_reference_vectors(native, P):
    """The known-answer vectors the reference's own numerics.py produced (the ones
    test_oracle_numerics.py pins the numpy restatement with)."""
    K = golden('numerics_kat.npz')
    t = 'P%d_' % P
    g = lambda n: K[t + n]
    A = g('hyper').shape[0]
    nm = native
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
        'fast_invert_nat_vi_delta': nm.fast_invert_nat_vi_delta(g('mu'), g('mu2'), g('const'),
                                                               g('natd')),
    }
    assert set(checks) == set(native.NATIVE)
    for name, got in checks.items():
        np.testing.assert_allclose(got, g(name), rtol=1e-11, atol=1e-13, err_msg=name)


def test_trajectory_with_native_numerics(native):
    """The oracle with the compiled loops enabled reproduces a reference-recorded trajectory."""
    g = golden('traj_p2_scale_se.npz')
    native.enable(threads=4)
    try:
        vi, _ = oracle_from_traj(g)
        np.random.seed(int(g['seed']))
        params = vi._initialize()
        elbo = vi.elbo(params)
        assert abs(elbo - float(g['init_elbo'])) < 1e-9 * abs(elbo)
        L, red = np.ones(5), None
        for it in range(len(g['elbo'])):
            params, L, elbo, red = vi._optimize_step(params, L, elbo, 2., red)
            assert abs(elbo - g['elbo'][it]) < 1e-9 * abs(elbo)
            assert np.array_equal(L, g['L'][it])
    finally:
        native.disable()
