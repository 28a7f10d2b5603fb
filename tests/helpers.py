"""Shared test helpers: golden loading and oracle construction (tests may import oracle/)."""
import contextlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TRAJ_NAMES = ['p1_dense', 'p2_lowrank', 'p2_scale_se', 'p4_general', 'p1_scaled', 'p4_m81',
              'p2_bigblock', 'p2_bigblock_lr']


@contextlib.contextmanager
def engine_class(cls):
    """TESTS ONLY.  The product has no parameter for its engine: MultiPopVI builds
    vilma_amd.engine.HipEngine(...), which needs a GPU.  The CPU tests of the host-side logic
    (driver loop, sharding, CLI plumbing) swap that module attribute for the oracle-backed test
    engine while they construct their objects, and put it back."""
    from vilma_amd import engine
    old = engine.HipEngine
    if cls is not None:
        engine.HipEngine = cls
    try:
        yield
    finally:
        engine.HipEngine = old


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def traj_blocks(g):
    """Dense per-cohort block matrices stored in a trajectory golden (a compact golden stores
    AR(1) LD as its parameter: R_ij = rho^|i-j|)."""
    P = int(g['P'])
    nb = len(g['sizes'])
    if 'ld_rho' in g.files:
        out = []
        for p in range(P):
            this = []
            for b, n in enumerate(g['sizes']):
                idx = np.arange(int(n))
                this.append(float(g['ld_rho'][p, b]) ** np.abs(idx[:, None] - idx[None, :]))
            out.append(this)
        return out
    return [[g['ld_%d_%d' % (p, b)] for b in range(nb)] for p in range(P)]


def oracle_from_traj(g, num_its=None):
    from oracle.ldop import EigenBlock, BlockDiagonalLD
    from oracle.vi import MultiPopVIOracle
    t = float(g['ldthresh'])
    ld = [BlockDiagonalLD([EigenBlock(X, t) for X in blocks], perm=g['perm'],
                          missing=g['missing'])
          for blocks in traj_blocks(g)]
    vi = MultiPopVIOracle(marginal_effects=g['betahat'], std_errs=g['se'], ld_mats=ld,
                          mixture_covs=list(g['covs']), annotations=g['annotations'],
                          checkpoint=False, checkpoint_freq=-1, output='t',
                          scaled=bool(g['scaled']), scale_se=bool(g['scale_se']),
                          gwas_N=g['gwas_N'], init_hg=g['init_hg'],
                          num_its=len(g['elbo']) if num_its is None else num_its)
    return vi, ld


def relerr(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300))) if a.size else 0.0


def engine_blocks(ld, form):
    """LD blocks of an oracle BlockDiagonalLD in the form the HIP engine takes."""
    out = []
    for b in ld.blocks:
        n, r = b.u.shape
        f = form
        if form == 'auto':
            f = 'dense' if 0.5 * n * n + 64.0 * n <= 2.0 * n * r else 'eig'
        if f == 'dense':
            out.append(('dense', (b.u * b.s) @ b.u.T))
        else:
            out.append(('eig', b.u, b.s))
    return out


def engine_from_oracle(vi, ld, form='auto'):
    """A HipEngine loaded with the same shard as the oracle object `vi`."""
    from vilma_amd.engine import HipEngine
    P, N, M, A = vi.num_pops, vi.num_loci, vi.num_mix, vi.num_annotations
    eng = HipEngine(P, N, M, A)
    eng.set_snp_data(vi.adj_marginal_effects, vi.std_errs, vi.scaled_ld_diags, vi.scalings,
                     vi.annotations)
    eng.set_mixture(vi.mixture_prec[:, :, :, 0], vi.log_det)
    eng.set_tau(vi.error_scaling)
    for p in range(P):
        n_ld = int(ld[p].starts[-1])
        eng.load_ld(p, engine_blocks(ld[p], form), ld[p].perm, n_ld)
    return eng


def oracle_totals(vi, params):
    """The VILMA_NTOTALS(P) = 3P+2 sums (include/vilma_hip.h) computed with the oracle."""
    from oracle import numerics as nm
    vi_mu, vi_delta, hyper = params
    mean = vi._posterior_mean(vi_mu, vi_delta)
    var = vi._posterior_marginal_variance(mean, vi_mu, vi_delta)
    z = mean / vi.std_errs
    linked = np.stack([vi.ld_mats[p].dot(z[p]) for p in range(vi.num_pops)])
    return np.concatenate([
        (mean * vi.adj_marginal_effects).sum(axis=1),
        (vi.scaled_ld_diags * var).sum(axis=1),
        (linked * z).sum(axis=1),
        [nm.fast_delta_kl(vi_delta, hyper, vi.annotations)
         + nm.fast_beta_kl(vi.sigma_summary, vi_delta),
         nm.fast_inner_product_comp(vi_mu, vi.mixture_prec, vi_delta)]])


def product_vi_from_traj(g, num_its=None, engine_factory=None, comm=None, form='auto'):
    """vilma_amd.MultiPopVI built from a golden problem through the product's own
    LowRankMatrix / BlockDiagonalMatrix (class-API depth of SURVEY.md section 8b)."""
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd.variational_inference import MultiPopVI
    t = float(g['ldthresh'])
    ld = [BlockDiagonalMatrix([LowRankMatrix(X, t) for X in blocks], perm=g['perm'],
                              missing=g['missing'])
          for blocks in traj_blocks(g)]
    with engine_class(engine_factory):
        vi = MultiPopVI(marginal_effects=g['betahat'], std_errs=g['se'], ld_mats=ld,
                        mixture_covs=list(g['covs']), annotations=g['annotations'],
                        checkpoint=False, checkpoint_freq=-1, output='t',
                        scaled=bool(g['scaled']), scale_se=bool(g['scale_se']),
                        gwas_N=g['gwas_N'], init_hg=g['init_hg'],
                        num_its=len(g['elbo']) if num_its is None else num_its, form=form,
                        _comm=comm)
    return vi, ld


def check_trajectory(vi, g, rtol_elbo=1e-9, rtol_mean=1e-7):
    """Drive `vi` (product class) through the golden's sweeps exactly as the reference loop
    does (variational_inference.py:353-389) and compare with what the reference recorded."""
    for key in ('ld_diags', 'adj_marginal_effects', 'chi_stat', 'ld_ranks', 'inverse_betas',
                'scalings', 'mixture_prec', 'log_det'):
        np.testing.assert_allclose(getattr(vi, key), g[key], rtol=1e-8, atol=1e-12, err_msg=key)
    np.random.seed(int(g['seed']))
    params = vi._initialize()
    elbo = vi.elbo(params)
    assert abs(elbo - float(g['init_elbo'])) < rtol_elbo * abs(elbo)
    np.testing.assert_allclose(params[2], g['init_hyper_delta'], rtol=1e-10)
    L = np.ones(5)
    red = None
    n_sweeps = len(g['elbo'])
    for it in range(n_sweeps):
        e0 = vi.n_evaluations
        params, L, elbo, red = vi._optimize_step(params, L=L, curr_elbo=elbo,
                                                 line_search_rate=2., running_elbo_delta=red)
        assert abs(elbo - g['elbo'][it]) < rtol_elbo * abs(elbo), (it, elbo, g['elbo'][it])
        assert np.array_equal(L, g['L'][it]), (it, L, g['L'][it])
        np.testing.assert_allclose(vi.error_scaling, g['error_scaling'][it], rtol=1e-8)
        np.testing.assert_allclose(params[2], g['hyper_delta'][it], rtol=1e-6, atol=1e-300)
        # one LD product per cohort per DISTINCT candidate point: the reference's count minus
        # its redundant re-evaluations (2 per sweep + 1 per inner beta step, +1 with scale_se)
        assert vi.n_evaluations - e0 <= int(g['objs_per_sweep'][it])
        if it in (0, n_sweeps - 1):
            np.testing.assert_allclose(vi.real_posterior_mean(params), g['post_mean'][it],
                                       rtol=rtol_mean, atol=1e-12)
    np.testing.assert_allclose(params[0], g['final_vi_mu'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(params[1], g['final_vi_delta'], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(vi.real_posterior_variance(params), g['final_post_var'],
                               rtol=1e-7)
    np.testing.assert_allclose(vi.vi_sigma, g['final_vi_sigma'], rtol=1e-9)
    return params


def check_compact_trajectory(vi, g, lookahead=False, rtol_elbo=1e-9, rtol_mean=1e-7):
    """check_trajectory for a compact (mid-size) golden: ELBO, L, running change and hyper_delta
    per sweep, posterior means at the recorded sweeps, final vi_mu and variance.  lookahead=True
    drives the sweeps through SweepDriver.sweep with the promise of a next call, i.e. through the
    device-resident path (sweeps queued ahead, decided on the device)."""
    for key in ('ld_diags', 'adj_marginal_effects', 'chi_stat', 'ld_ranks', 'inverse_betas'):
        np.testing.assert_allclose(getattr(vi, key), g[key], rtol=1e-8, atol=1e-12, err_msg=key)
    np.random.seed(int(g['seed']))
    params = vi._initialize()
    elbo = vi.elbo(params)
    assert abs(elbo - float(g['init_elbo'])) < rtol_elbo * abs(elbo)
    np.testing.assert_allclose(params[2], g['init_hyper_delta'], rtol=1e-10)
    n_sweeps = len(g['elbo'])
    marks = {int(s): i for i, s in enumerate(g['post_mean_sweeps'])}
    state, L, red = None, np.ones(5), None
    counts = []
    for it in range(n_sweeps):
        e0, t0 = vi.n_evaluations, vi.n_trials
        if lookahead:
            state, _ = vi.sweep(state, lookahead=it + 1 < n_sweeps)
            L, elbo, red = state['L'], state['elbo'], state['running']
            params = vi._params()
        else:
            params, L, elbo, red = vi._optimize_step(params, L=L, curr_elbo=elbo,
                                                     line_search_rate=2., running_elbo_delta=red)
        counts.append((vi.n_evaluations - e0, vi.n_trials - t0))
        assert abs(elbo - g['elbo'][it]) < rtol_elbo * abs(elbo), (it, elbo, g['elbo'][it])
        assert np.array_equal(L, g['L'][it]), (it, L, g['L'][it])
        assert abs(red - g['running_elbo_delta'][it]) <= 1e-7 * abs(red) + 1e-9, (it, red)
        if it in marks:
            np.testing.assert_allclose(vi.real_posterior_mean(params), g['post_mean'][marks[it]],
                                       rtol=rtol_mean, atol=1e-12)
            np.testing.assert_allclose(params[2], g['hyper_delta'][it], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(params[0], g['final_vi_mu'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(vi.real_posterior_variance(params), g['final_post_var'], rtol=1e-7)
    return counts
