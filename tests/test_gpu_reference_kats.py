"""The behaviours the reference's own tests pin (/root/reference/tests/test.py:28-477, 1297-1411,
1473-1514, 1582-1604, 1704-1726, 1849-1876), asserted on the PRODUCT classes running on the GPU:

  * vilma_amd.matrix_structures.BlockDiagonalMatrix.dot (device) and what composes with it, against
    the vectors the reference's classes produced (tests/golden/ldop_kat.npz);
  * vilma_amd.MultiPopVI on the two small problems the reference's tests are written on: the
    closed forms of test_MultiPopVI_init, and the outputs of the reference's _initialize,
    _nat_to_not_vi_delta, _update_error_scaling, _update_beta (twice: idempotence),
    _update_hyper_delta, _nat_grad_step and optimize() recorded in tests/golden/vischeme_kat.npz by
    make_golden.py.
"""
import numpy as np
import pytest

from helpers import golden

pytestmark = pytest.mark.gpu

K = golden('ldop_kat.npz')
V = golden('vischeme_kat.npz')


# ---------------------------------------------------------------------------------------------
# matrix_structures on the device
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('t', [1.0, 0.8, 0.3])
def test_block_diagonal_dot_on_the_device_against_the_reference_vectors(t):
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    tag = 't%02d_' % int(t * 10)
    bd = BlockDiagonalMatrix([LowRankMatrix(K['X%d' % b], t) for b in range(3)], perm=K['perm'],
                             missing=K['missing'])
    vec = K['vec']
    got = bd.dot(vec)
    np.testing.assert_allclose(got, K[tag + 'dot'], atol=1e-12)
    assert np.all(got[K['missing']] == 0)
    np.testing.assert_allclose(bd.dot(np.stack([vec, 2 * vec - 1], axis=1)), K[tag + 'dot_matrix'],
                               atol=1e-12)
    # row i of the operator, one at a time, is the same product
    np.testing.assert_allclose([bd.dot_i(vec, i) for i in range(len(vec))], got, atol=1e-12)
    # matrix_power(0.5) (identity perm, as the reference builds it) and its square
    half = bd.matrix_power(0.5)
    np.testing.assert_allclose(half.dot(vec), K[tag + 'sqrt_dot'], atol=1e-12)
    # pseudo-inverse of the pseudo-inverse is the operator again
    np.testing.assert_allclose(bd.inverse.inverse.dot(vec), K[tag + 'inv_inv_dot'], atol=1e-12)
    # R R^+ R x = R x
    np.testing.assert_allclose(bd.dot(bd.inverse.dot(got)), got, atol=1e-9)
    # ridge solve inverts (R + diag(reg)) on the covered SNPs
    y = bd.ridge_inverse_dot(vec, K['reg'])
    cov = np.setdiff1d(np.arange(len(vec)), K['missing'])
    np.testing.assert_allclose((bd.dot(y) + K['reg'] * y)[cov], vec[cov], rtol=1e-9, atol=1e-11)


def test_degenerate_block_on_the_device():
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    bd = BlockDiagonalMatrix([LowRankMatrix(K['degenerate_X'], 0.5)])
    assert np.array_equal(bd.dot(np.arange(4.0)), K['degenerate_dot'])
    assert bd.get_rank() == 0


# ---------------------------------------------------------------------------------------------
# MultiPopVI on the reference tests' own problems
# ---------------------------------------------------------------------------------------------
def _problem(linked, num_annotations):
    """2 cohorts x 50 SNPs, one LD block, two mixture components (the numbers of
    tests/test.py:1225-1293; make_golden.vischeme_problem is the generator's twin)."""
    if linked:
        betas = np.arange(100).reshape(2, 50).astype(float)
        ld = (1 + np.arange(50 * 50)).reshape(50, 50) / (50 * 50 + 1)
        ld = ld + ld.T + 5 * np.eye(50)
        d = np.diag(1 / np.sqrt(np.diag(ld)))
        ld = d @ ld @ d
    else:
        betas = np.arange(100).reshape(50, 2).T.astype(float)
        ld = np.eye(50)
    std_errs = np.array([1.] * 50 + [2.] * 50).reshape(2, 50)
    ann = np.ones((50, 1), dtype=int)
    if num_annotations == 2:
        ann = np.zeros((50, 2), dtype=int)
        ann[0:25, 0] = 1
        ann[25:, 1] = 1
    return betas, std_errs, ld, ann


def _vischeme(linked, num_annotations, scaled, scale_se, num_its=20):
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd.variational_inference import MultiPopVI
    betas, std_errs, ld, ann = _problem(linked, num_annotations)
    lr = LowRankMatrix(X=ld, t=1.0)
    return MultiPopVI(marginal_effects=betas, std_errs=std_errs,
                      ld_mats=[BlockDiagonalMatrix([lr]), BlockDiagonalMatrix([lr])],
                      mixture_covs=[np.eye(2), 2 * np.eye(2)], annotations=ann, checkpoint=False,
                      checkpoint_freq=-1, output='test', scaled=scaled, scale_se=scale_se,
                      gwas_N=np.array([100e3, 10e3]), init_hg=np.array([0.1, 0.9]), num_its=num_its)


CASES = {'linked_a2': (True, 2, False, False), 'unlinked_a1': (False, 1, False, False),
         'linked_a1_scaled': (True, 1, True, False), 'linked_a2_scale_se': (True, 2, False, True),
         'linked_a2_scaled_scale_se': (True, 2, True, True)}


def test_init_closed_forms():
    """test_MultiPopVI_init: sigma-dependent constants in closed form, attributes, the load-time
    constants against dense linear algebra."""
    vi = _vischeme(True, 2, False, False)
    betas, std_errs, ld, _ = _problem(True, 2)
    assert vi.num_pops == 2 and vi.num_mix == 2
    assert set(vi.param_names) == {'vi_mu', 'vi_delta', 'hyper_delta'}
    assert vi.mixture_prec.shape == (2, 2, 2, 1)
    np.testing.assert_allclose(vi.mixture_prec[0, :, :, 0], np.eye(2))
    np.testing.assert_allclose(vi.mixture_prec[1, :, :, 0], 0.5 * np.eye(2))
    np.testing.assert_allclose(vi.log_det, [0., 2 * np.log(2)])
    # vi_sigma = (prec_k + diag(1 / se_p^2))^-1 : 1/(1+1), 1/(1+1/4), 1/(1/2+1), 1/(1/2+1/4)
    sig = np.zeros((2, 2, 2, 50))
    sig[0, 0, 0], sig[0, 1, 1], sig[1, 0, 0], sig[1, 1, 1] = 1 / 2, 4 / 5, 2 / 3, 4 / 3
    nat = np.zeros((2, 2, 2, 50))
    nat[0, 0, 0], nat[0, 1, 1], nat[1, 0, 0], nat[1, 1, 1] = -1, -5 / 8, -3 / 4, -3 / 8
    np.testing.assert_allclose(vi.vi_sigma, sig)
    np.testing.assert_allclose(vi.nat_sigma, nat)
    logdet = np.zeros((2, 50))
    logdet[0], logdet[1] = np.log(2 / 5), np.log(8 / 9)
    np.testing.assert_allclose(vi.vi_sigma_log_det, logdet)
    matches = np.zeros((50, 2))
    matches[:, 0], matches[:, 1] = 1 / 2 + 4 / 5, 1 / 3 + 2 / 3
    np.testing.assert_allclose(vi.vi_sigma_matches, matches)
    np.testing.assert_allclose(vi.sigma_summary,
                               np.array([0., 2 * np.log(2)]) - logdet.T + matches)
    assert vi.nat_grad_vi_delta is None
    assert not vi.scaled and not vi.scale_se and np.allclose(vi.error_scaling, 1)
    assert vi.num_annotations == 2 and np.allclose(vi.annotation_counts, 25)
    assert np.all(vi.annotations[:25] == 0) and np.all(vi.annotations[25:] == 1)
    np.testing.assert_allclose(vi.marginal_effects, betas)
    np.testing.assert_allclose(vi.std_errs, std_errs)
    assert np.allclose(vi.scalings, 1) and np.allclose(vi.ld_diags, 1)
    np.testing.assert_allclose(vi.scaled_ld_diags, std_errs ** -2)
    assert len(vi.ld_mats) == 2 and vi.ld_mats[0].shape == (50, 50)
    assert vi.checkpoint_freq == -1 and vi.checkpoint_path == 'test-checkpoint'
    np.testing.assert_allclose(vi.init_hg, [0.1, 0.9])
    np.testing.assert_allclose(vi.gwas_N, [100e3, 10e3])
    assert vi.num_its == 20
    inv = np.linalg.inv(ld)
    for p in range(2):
        z = betas[p] / std_errs[p]
        np.testing.assert_allclose(vi.adj_marginal_effects[p], inv.dot(ld.dot(z)) / std_errs[p],
                                   rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(vi.chi_stat[p], betas[p].dot(inv.dot(z) / std_errs[p]), rtol=1e-8)
    assert np.allclose(vi.ld_ranks, 50)
    prior = 2 * np.array([100e3, 10e3]) * np.array([0.1, 0.9]) / (std_errs ** -2).sum(axis=1)
    temp = ld.dot(inv.dot((betas / std_errs).T)).T
    for p in range(2):
        want = np.linalg.inv(ld + np.diag(std_errs[p] ** 2) / prior[p]).dot(temp[p]) * std_errs[p]
        np.testing.assert_allclose(vi.inverse_betas[p], want, rtol=1e-7, atol=1e-10)
    vi.engine.close()


@pytest.mark.parametrize('case', sorted(CASES))
def test_constants_and_private_steps_against_the_reference(case):
    linked, A, scaled, scale_se = CASES[case]
    t = case + '_'
    vi = _vischeme(linked, A, scaled, scale_se)
    for key in ('vi_sigma', 'nat_sigma', 'vi_sigma_log_det', 'vi_sigma_matches', 'sigma_summary',
                'adj_marginal_effects', 'chi_stat', 'ld_ranks', 'inverse_betas', 'scaled_ld_diags',
                'ld_diags', 'scalings', 'log_det'):
        np.testing.assert_allclose(getattr(vi, key), V[t + key], rtol=1e-8, atol=1e-10, err_msg=key)
    # _initialize: same RNG stream, same starting point
    np.random.seed(42)
    params = vi._initialize()
    np.testing.assert_allclose(params[0], V[t + 'init_mu'], rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(params[2], V[t + 'init_hyper'], rtol=1e-10)
    # the reference's _initialize keeps the HEURISTIC responsibilities as vi_delta; elbo() and the
    # posterior moments of exactly that tuple (the vi_delta as given, not the fixed point)
    given = (V[t + 'init_mu'], V[t + 'init_delta'], V[t + 'init_hyper'])
    assert abs(vi.elbo(given) - float(V[t + 'init_elbo'])) < 1e-9 * abs(float(V[t + 'init_elbo']))
    np.testing.assert_allclose(vi.real_posterior_mean(*given), V[t + 'init_post_mean'], rtol=1e-8,
                               atol=1e-13)
    np.testing.assert_allclose(vi.real_posterior_variance(*given), V[t + 'init_post_var'], rtol=1e-8)
    # _nat_to_not_vi_delta (test_MultiPopVI_nat_to_not_vi_delta)
    fp = vi._nat_to_not_vi_delta(given)
    np.testing.assert_allclose(fp[1], V[t + 'fixed_point_delta'], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(fp[0], given[0], rtol=1e-12)
    np.testing.assert_allclose(fp[2], given[2], rtol=1e-12)
    # _update_error_scaling at that point (test_MultiPopVI_update_error_scaling): the reference
    # evaluates the moments at the vi_delta it is handed -- the heuristic one of _initialize
    post_mean = vi._posterior_mean(*given)
    pmv = vi._posterior_marginal_variance(post_mean, *given)
    true_tau = np.zeros(2)
    for p in range(2):
        true_tau[p] = 1. / vi.ld_ranks[p] * (
            vi.chi_stat[p] - 2 * vi.adj_marginal_effects[p].dot(post_mean[p])
            + post_mean[p].dot(vi.ld_mats[p].dot(post_mean[p] / vi.std_errs[p]) / vi.std_errs[p])
            + (vi.scaled_ld_diags[p] * pmv[p]).sum())
    np.testing.assert_allclose(true_tau, V[t + 'tau_after_update'], rtol=1e-8)
    vi.error_scaling = np.ones(2)
    vi._set_vi_sigma()
    # _update_beta twice from the fixed point (test_MultiPopVI_update_beta: with unlinked LD the full
    # natural-gradient step is accepted at L = 1, raises the objective, and a second one changes nothing)
    p1, L1, o1, n1 = vi._update_beta(*fp, None, [1., 1., 1.], 0, 1.25)
    np.testing.assert_allclose([o1, n1], V[t + 'ub1_objs'], rtol=1e-9)
    np.testing.assert_allclose(L1, V[t + 'ub1_L'])
    np.testing.assert_allclose(p1[0], V[t + 'ub1_mu'], rtol=1e-7, atol=1e-13)
    np.testing.assert_allclose(p1[1], V[t + 'ub1_delta'], rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose(p1[2], given[2], rtol=1e-12)
    assert n1 >= o1 - 1e-6 * abs(o1) - 1e-6
    p2, L2, o2, n2 = vi._update_beta(*p1, None, [1., 1., 1.], 0, 1.25)
    np.testing.assert_allclose([o2, n2], V[t + 'ub2_objs'], rtol=1e-9)
    np.testing.assert_allclose(p2[0], V[t + 'ub2_mu'], rtol=1e-7, atol=1e-13)
    if not linked:
        assert L1[0] == 1 and n1 > o1
        assert np.isclose(n2, o2) and np.allclose(p2[0], p1[0]) and np.allclose(p2[1], p1[1])
    # _update_hyper_delta from p1
    p3, _, o3, n3 = vi._update_hyper_delta(*p1, None, [1., 1., 1.], 1, 1.25)
    np.testing.assert_allclose(p3[2], V[t + 'uh_hyper'], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(p3[1], V[t + 'uh_delta'], rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose([o3, n3], V[t + 'uh_objs'], rtol=1e-9)
    # _nat_grad_step from the fixed point (test_MultiPopVI_nat_grad_step)
    vi.error_scaling = np.ones(2)
    vi._set_vi_sigma()
    q, Lq, dq = vi._nat_grad_step(fp, [1., 1., 1.], 2., None)
    np.testing.assert_allclose(Lq[:3], V[t + 'ngs_L'])
    assert abs(dq - float(V[t + 'ngs_delta_elbo'])) < 1e-8 * abs(dq) + 1e-9
    np.testing.assert_allclose(q[0], V[t + 'ngs_mu'], rtol=1e-7, atol=1e-13)
    np.testing.assert_allclose(q[2], V[t + 'ngs_hyper'], rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose(vi.error_scaling, V[t + 'ngs_tau'], rtol=1e-8)
    assert vi.elbo(q) > vi.elbo(fp)
    vi.engine.close()


@pytest.mark.parametrize('case', sorted(CASES))
def test_optimize_raises_the_elbo_and_ends_where_the_reference_ends(case):
    """test_MultiPopVI_optimize (monotone ELBO) plus the end point the reference reaches from the
    same seed in its 20 iterations."""
    linked, A, scaled, scale_se = CASES[case]
    t = case + '_'
    vi = _vischeme(linked, A, scaled, scale_se)
    np.random.seed(42)
    start = vi._initialize()
    e0 = vi.elbo((start[0], start[1], start[2]))
    np.random.seed(42)
    final = vi.optimize()
    e1 = vi.elbo(final)
    assert e1 > e0
    assert abs(e1 - float(V[t + 'opt_elbo'])) < 1e-8 * abs(e1)
    np.testing.assert_allclose(vi.real_posterior_mean(final), V[t + 'opt_post_mean'], rtol=1e-6,
                               atol=1e-11)
    np.testing.assert_allclose(vi.error_scaling, V[t + 'opt_tau'], rtol=1e-8)
    np.testing.assert_allclose(final[2], V[t + 'opt_hyper'], rtol=1e-6, atol=1e-300)
    vi.engine.close()


def test_nat_grad_step_backs_off_from_a_huge_L():
    """test_MultiPopVI_nat_grad_step, second half: starting the line search next to L_MAX the step
    is tiny (vi_mu does not move to working precision), L comes down, the ELBO still rises through
    the M-step."""
    vi = _vischeme(False, 1, False, False)
    np.random.seed(42)
    start = vi._initialize()
    fp = vi._nat_to_not_vi_delta((start[0], start[1], start[2]))
    q, Lq, _ = vi._nat_grad_step(fp, [1e12 - 1, 1., 1.], 2., None)
    assert Lq[0] < 1e12 - 1
    assert vi.elbo(q) > vi.elbo(fp)
    np.testing.assert_allclose(q[0], fp[0], rtol=1e-5, atol=1e-8)
    vi.engine.close()
