"""GPU tests of the load-time path: blocks stream host -> device as they
are decomposed; diag(R), R^+ z, chi, R R^+ z per block and the ridge start come from the device
and must equal what the reference's per-block host formulas give (oracle/ldop.py,
matrix_structures.py:159-196, 349-387, 426-447)."""
import os

import numpy as np
import pytest

from helpers import golden, oracle_from_traj, product_vi_from_traj, GOLDEN

pytestmark = pytest.mark.gpu


def _rank_deficient(rng, n, r):
    """A correlation-like PSD matrix of rank r < n (what a reference panel with few samples gives)."""
    f = rng.normal(size=(n, r))
    c = f @ f.T
    d = 1.0 / np.sqrt(np.diag(c))
    return c * np.outer(d, d)


@pytest.mark.parametrize('gpu_eigh', ['planned', 'all', 'none'])
@pytest.mark.parametrize('form', ['auto', 'dense', 'eig'])
def test_device_loader_matches_host_formulas(form, gpu_eigh, monkeypatch):
    from oracle.ldop import EigenBlock, BlockDiagonalLD
    from vilma_amd.engine import HipEngine
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd import ld_device
    # where eigh runs must not matter: the cost model's choice, every block on the GPU
    # (rocSOLVER), every block on the host pool (LAPACK)
    if gpu_eigh == 'all':
        monkeypatch.setattr(ld_device, 'plan_gpu_eigh',
                            lambda sizes, deferred, workers: {b for b, d in enumerate(deferred) if d})
    elif gpu_eigh == 'none':
        monkeypatch.setenv('VILMA_GPU_EIGH', '0')
    rng = np.random.default_rng(11)
    sizes = [1, 40, 257, 130, 600, 2]
    mats = [np.ones((1, 1)), _rank_deficient(rng, 40, 12), _rank_deficient(rng, 257, 100),
            0.7 ** np.abs(np.subtract.outer(np.arange(130), np.arange(130))),
            _rank_deficient(rng, 600, 480), np.array([[1.0, 0.3], [0.3, 1.0]])]
    n_ld = sum(sizes)
    N = n_ld + 7
    order = rng.permutation(N)
    perm = np.concatenate([order[:n_ld], np.sort(order[n_ld:])]).astype(np.int64)
    for t in (1.0, 0.6):
        made = []

        def thunk(X):
            def make():
                made.append(X.shape[0])
                return X
            return make
        ld = BlockDiagonalMatrix([LowRankMatrix.deferred(thunk(X), X.shape[0], t) for X in mats],
                                 perm=perm, missing=perm[n_ld:])
        old = BlockDiagonalLD([EigenBlock(X, t) for X in mats], perm=perm, missing=perm[n_ld:])
        z = rng.normal(size=N)
        z[perm[n_ld:]] = 0.0
        eng = HipEngine(1, N, 2, 1)
        out = ld_device.stream_cohort(eng, 0, ld, form, z[perm[:n_ld]], workers=3)
        assert out['gpu_eigh'] == {'all': len(sizes), 'none': 0}.get(gpu_eigh, out['gpu_eigh'])
        assert sorted(made) == sorted(sizes)                     # every block decomposed exactly once
        assert all(m.is_deferred() for m in ld.matrices)        # ... and dropped from host memory
        diag, rmle = np.zeros(N), np.zeros(N)
        diag[perm[:n_ld]], rmle[perm[:n_ld]] = out['diag'], out['rmle']
        mle = old.inverse_dot(z)
        np.testing.assert_allclose(diag, old.diag(), rtol=1e-11, atol=1e-13)
        assert out['rank'] == old.get_rank()
        np.testing.assert_allclose(out['chi'], z.dot(mle), rtol=1e-10)
        np.testing.assert_allclose(rmle, old.dot(mle), rtol=1e-9, atol=1e-11)
        x = rng.normal(size=N)
        np.testing.assert_allclose(eng.ld_matvec(x[None])[0], old.dot(x), rtol=1e-10, atol=1e-11)
        # ridge start: well conditioned (typical) and badly conditioned (tiny regulariser)
        # (condition number ~1e7: the reference's explicit r x r inverse and the iteration both
        # carry ~1e-9 relative error there)
        for scale, tol in ((30.0, 1e-11), (1e-4, 1e-7)):
            reg = scale * rng.uniform(0.5, 2.0, size=N)
            want = old.ridge_inverse_dot(rmle, reg)
            got = ld_device.ridge_start(eng, rmle[None], reg[None], diag[None])[0]
            np.testing.assert_allclose(got, want, rtol=1e-8, atol=tol * np.abs(want).max())
        # an iteration budget too small for 1e-13: the iterate is accepted (with a warning) once its
        # TRUE residual is below 1e-6 -- it only seeds a starting point -- and raises RidgeStalled,
        # which MultiPopVI answers with the reference's per-block solve, when not even that is reached
        reg = 30.0 * rng.uniform(0.5, 2.0, size=N)
        want = old.ridge_inverse_dot(rmle, reg)
        got = ld_device.ridge_start(eng, rmle[None], reg[None], diag[None], max_iter=8)[0]
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * np.abs(want).max())
        with pytest.raises(ld_device.RidgeStalled):
            ld_device.ridge_start(eng, rmle[None], 1e-7 * reg[None], diag[None], max_iter=8)
        eng.close()


def test_gpu_eigh_keeps_the_ranks_lapack_keeps():
    """Kept ranks and reconstructions of rocSOLVER's eigh on the GPU against LAPACK's on the host,
    on blocks of the C3 synthetic law (AR(1), full rank), on rank-deficient panels and under
    --ldthresh 0.8, incl. the degenerate selections (nothing kept)."""
    import torch
    from vilma_amd import ld_device
    from vilma_amd.matrix_structures import LowRankMatrix
    from vilma_amd.synthetic import ar1_numpy
    dev = torch.device('cuda', 0)
    staging = ld_device._Staging(torch, dev)
    compute = torch.cuda.current_stream(dev)
    rng = np.random.default_rng(4)
    cases = [(ar1_numpy(n, rho), t) for n, rho, t in
             ((75, 0.5, 1.0), (588, 0.93, 1.0), (1300, 0.7, 1.0), (2431, 0.95, 1.0), (700, 0.9, 0.8))]
    cases += [(_rank_deficient(rng, 900, 300), 1.0), (_rank_deficient(rng, 640, 480), 0.8),
              (-np.eye(5), 1.0), (np.zeros((4, 4)), 1.0), (0.05 * np.eye(6), 0.5)]
    def check(X, t, got):
        host = LowRankMatrix(X, t)
        Ud, sd, s = got
        assert Ud.shape == host.u.shape and sd.shape == host.s.shape, (X.shape, t)
        np.testing.assert_allclose(s, host.s, rtol=1e-10, atol=1e-12)
        recon = ((Ud * sd) @ Ud.T).cpu().numpy()
        np.testing.assert_allclose(recon, host.reconstruct(), rtol=0, atol=1e-10)
    for X, t in cases:
        check(X, t, ld_device._gpu_factors(torch, staging, compute, X, t))
    # the same blocks in stacks: ragged sizes padded to a common one (the padding's eigenvalue
    # sits below each block's Gershgorin bound, also for the negative-definite case)
    small = [(k, X, t) for k, (X, t) in enumerate(cases) if X.shape[0] <= 128]
    mid = [(k, X, t) for k, (X, t) in enumerate(cases) if 576 < X.shape[0] <= 704]
    assert len(small) >= 4 and len(mid) >= 3
    for group in (small, mid):
        got = ld_device._gpu_factors_stacked(torch, staging, compute, group)
        for k, X, t in group:
            check(X, t, got[k])


def test_lazy_schema_streams_to_the_device(tmp_path):
    """The CLI's lazy schema loading (load.py) feeds the streaming loader: same constants and
    the same fit as eager loading, and no block's factors are left on the host afterwards."""
    from vilma_amd import load
    from vilma_amd.variational_inference import MultiPopVI
    ref = os.path.join(GOLDEN, 'refdata')
    variants = load.load_variant_list(os.path.join(ref, 'good_variants.tsv'))
    stats, _ = load.load_sumstats(os.path.join(ref, 'good_sumstats_beta.tsv'), variants)
    beta, se = stats.BETA.to_numpy()[None], stats.SE.to_numpy()[None]
    covs = [np.array([[v]]) for v in np.geomspace(1e-6, 1e-1, 8)]
    fits = {}
    for lazy in (False, True):
        ld, _ = load.load_ld_from_schema(os.path.join(ref, 'ld_manifest.tsv'), variants, [], 0.8,
                                         lazy=lazy)
        vi = MultiPopVI(marginal_effects=beta, std_errs=se, ld_mats=[ld],
                        annotations=np.ones((beta.shape[1], 1)), mixture_covs=covs,
                        checkpoint=False, gwas_N=np.array([1e4]), init_hg=np.array([0.2]),
                        num_its=5)
        if lazy:
            assert all(m.is_deferred() for m in ld.matrices)
        np.random.seed(3)
        params = vi.optimize()
        fits[lazy] = (vi.ld_diags, vi.adj_marginal_effects, vi.chi_stat, vi.ld_ranks,
                      vi.inverse_betas, vi.real_posterior_mean(params))
        vi.engine.close()
    for a, b in zip(fits[False], fits[True]):
        np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-14)


def test_ridge_start_fallback_end_to_end(monkeypatch, caplog):
    """When conjugate gradients cannot deliver the ridge start (RidgeStalled), MultiPopVI answers
    with the reference's per-block solve on the host (matrix_structures.py:349-387) -- through
    blocks the streaming loader has already dropped from host memory (they are decomposed again)
    and blocks given as factors alike -- and ends with the reference's inverse_betas."""
    import logging
    from helpers import golden, traj_blocks
    from vilma_amd import ld_device
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd.variational_inference import MultiPopVI
    g = golden('traj_p2_lowrank.npz')
    t = float(g['ldthresh'])
    made = []

    def thunk(X):
        def make():
            made.append(X.shape[0])
            return X
        return make
    ld = []
    for p, blocks in enumerate(traj_blocks(g)):
        mats = []
        for b, X in enumerate(blocks):
            if b % 2 == 0:          # decomposed lazily (and forgotten once in HBM) ...
                mats.append(LowRankMatrix.deferred(thunk(X), X.shape[0], t))
            else:                   # ... or handed over as factors
                m = LowRankMatrix(X, t)
                mats.append(LowRankMatrix(u=m.u, s=m.s, v=m.v, D=np.zeros(X.shape[0])))
        ld.append(BlockDiagonalMatrix(mats, perm=g['perm'], missing=g['missing']))

    def stalled(*a, **k):
        raise ld_device.RidgeStalled('ridge start: forced by the test')
    monkeypatch.setattr(ld_device, 'ridge_start', stalled)
    with caplog.at_level(logging.WARNING):
        vi = MultiPopVI(marginal_effects=g['betahat'], std_errs=g['se'], ld_mats=ld,
                        mixture_covs=list(g['covs']), annotations=g['annotations'],
                        checkpoint=False, scaled=bool(g['scaled']), scale_se=bool(g['scale_se']),
                        gwas_N=g['gwas_N'], init_hg=g['init_hg'], num_its=3)
    assert any('falling back to the per-block ridge solve' in r.getMessage() for r in caplog.records)
    n_lazy = sum((len(blocks) + 1) // 2 for blocks in traj_blocks(g))
    assert len(made) == 2 * n_lazy          # once for the device store, once more for the fallback
    np.testing.assert_allclose(vi.inverse_betas, g['inverse_betas'], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(vi.adj_marginal_effects, g['adj_marginal_effects'], rtol=1e-8, atol=1e-12)
    np.random.seed(int(g['seed']))
    params = vi._initialize()
    assert abs(vi.elbo(params) - float(g['init_elbo'])) < 1e-9 * abs(float(g['init_elbo']))
    vi.engine.close()
