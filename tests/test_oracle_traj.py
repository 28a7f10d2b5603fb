"""Oracle VI engine vs trajectories recorded from the reference's MultiPopVI
(driven through _optimize_step exactly as variational_inference.py:353-389 does)."""
import numpy as np
import pytest

from helpers import golden, oracle_from_traj, relerr, TRAJ_NAMES


@pytest.mark.parametrize('name', TRAJ_NAMES)
def test_trajectory(name):
    g = golden('traj_%s.npz' % name)
    vi, ld = oracle_from_traj(g)
    # SNP -> block assignment is bit exact
    for p in range(int(g['P'])):
        assert np.array_equal(ld[p].perm, g['perm'])
        assert np.array_equal(ld[p].missing, g['missing'])
        for b, blk in enumerate(ld[p].blocks):
            assert blk.s.shape[0] == int(g['rank_%d_%d' % (p, b)])
    for key in ('ld_diags', 'adj_marginal_effects', 'chi_stat', 'ld_ranks', 'inverse_betas',
                'scalings', 'mixture_prec', 'log_det'):
        np.testing.assert_allclose(getattr(vi, key), g[key], rtol=1e-8, atol=1e-12, err_msg=key)
    np.random.seed(int(g['seed']))
    params = vi._initialize()
    assert relerr(params[0], g['init_vi_mu']) < 1e-8
    np.testing.assert_allclose(params[1], g['init_vi_delta'], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(params[2], g['init_hyper_delta'], rtol=1e-10)
    elbo = vi.elbo(params)
    assert abs(elbo - float(g['init_elbo'])) < 1e-9 * abs(elbo)
    L = np.ones(5)
    red = None
    for it in range(len(g['elbo'])):
        m0, o0 = vi.n_matvec, vi.n_objective
        params, L, elbo, red = vi._optimize_step(params, L=L, curr_elbo=elbo,
                                                 line_search_rate=2., running_elbo_delta=red)
        params = tuple(params)
        assert abs(elbo - g['elbo'][it]) < 1e-9 * abs(elbo), (it, elbo, g['elbo'][it])
        assert np.array_equal(L, g['L'][it]), it
        assert abs(red - g['running_elbo_delta'][it]) <= 1e-7 * abs(g['running_elbo_delta'][it]) + 1e-9
        # same schedule: matvec and objective-evaluation counts per sweep
        assert vi.n_matvec - m0 == int(g['dots_per_sweep'][it])
        assert vi.n_objective - o0 == int(g['objs_per_sweep'][it])
        np.testing.assert_allclose(vi.error_scaling, g['error_scaling'][it], rtol=1e-9)
        np.testing.assert_allclose(params[2], g['hyper_delta'][it], rtol=1e-7, atol=1e-300)
        np.testing.assert_allclose(vi.real_posterior_mean(*params), g['post_mean'][it],
                                   rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(params[0], g['final_vi_mu'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(params[1], g['final_vi_delta'], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(vi.real_posterior_variance(*params), g['final_post_var'], rtol=1e-7)
    np.testing.assert_allclose(vi.vi_sigma, g['final_vi_sigma'], rtol=1e-9)


@pytest.mark.parametrize('name', ['p1_dense', 'p2_scale_se'])
def test_optimize_convergence(name):
    g = golden('traj_%s.npz' % name)
    cap = {'p1_dense': 40, 'p2_scale_se': 30}[name]
    vi, _ = oracle_from_traj(g, num_its=cap)
    np.random.seed(int(g['seed']))
    params = vi.optimize()
    assert vi.num_its_run == int(g['opt_num_its'])
    np.testing.assert_allclose(vi.real_posterior_mean(*params), g['opt_post_mean'], rtol=1e-6,
                               atol=1e-12)
    np.testing.assert_allclose(vi.error_scaling, g['opt_error_scaling'], rtol=1e-8)


def test_mid_size_trajectory_with_inner_loops():
    """The compact mid-size golden (5 000 LD SNPs, 2 cohorts, 16 sweeps recorded from the
    reference): sweeps with several beta updates and a trial whose first two steps are both
    rejected -- the schedule (objective evaluations per sweep) as well as the values."""
    g = golden('traj_p2_mid.npz')
    assert int(g['objs_per_sweep'].max()) >= 9          # a sweep with an inner beta loop
    assert g['L'][0, 0] >= 4.0                           # 1 and 2 rejected in the very first update
    vi, ld = oracle_from_traj(g)
    for p in range(int(g['P'])):
        for b, blk in enumerate(ld[p].blocks):
            assert blk.s.shape[0] == int(g['rank_%d_%d' % (p, b)])
    np.random.seed(int(g['seed']))
    params = vi._initialize()
    elbo = vi.elbo(params)
    assert abs(elbo - float(g['init_elbo'])) < 1e-9 * abs(elbo)
    L, red = np.ones(5), None
    marks = {int(s): i for i, s in enumerate(g['post_mean_sweeps'])}
    for it in range(len(g['elbo'])):
        o0 = vi.n_objective
        params, L, elbo, red = vi._optimize_step(params, L=L, curr_elbo=elbo,
                                                 line_search_rate=2., running_elbo_delta=red)
        params = tuple(params)
        assert abs(elbo - g['elbo'][it]) < 1e-9 * abs(elbo), (it, elbo, g['elbo'][it])
        assert np.array_equal(L, g['L'][it]), it
        assert vi.n_objective - o0 == int(g['objs_per_sweep'][it])
        if it in marks:
            np.testing.assert_allclose(vi.real_posterior_mean(*params), g['post_mean'][marks[it]],
                                       rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(params[0], g['final_vi_mu'], rtol=1e-6, atol=1e-12)
