"""GPU tests at BASELINE.json's full sizes.

C2 (1 cohort, 100k SNPs, 500x200 blocks, M=25) is small enough for a direct comparison with the
oracle.  C3 (2 cohorts, 1M SNPs, ~1700 blocks, M=40) is checked through size-independent
properties: the LD operator against the O(n) closed form of an AR(1) product, linearity,
symmetry; the fit through ELBO monotonicity, line-search invariants and finiteness."""
import numpy as np
import pytest
from scipy.signal import lfilter

pytestmark = pytest.mark.gpu


def _setup(workload, seed=0, form='auto', block_range=None, scale_se=False):
    import torch
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS
    from vilma_amd.engine import HipEngine
    from vilma_amd.sharding import Comm
    from vilma_amd.variational_inference import SweepDriver
    device = torch.device('cuda', 0)
    sh = SyntheticShard(seed=seed, block_range=block_range, **WORKLOADS[workload]).build(device)
    sh.finish_init(sh.inv_se2_local)
    eng = HipEngine(sh.P, sh.N, sh.M, 1)
    eng.set_snp_data(sh.adj, sh.se, sh.sld, sh.scalings, sh.annot)
    prec, log_det = np.linalg.inv(sh.covs), np.linalg.slogdet(sh.covs)[1]
    eng.set_mixture(prec, log_det)
    for p in range(sh.P):
        if sh.kind == 'lowrank':
            eng.load_ld(p, sh.ld_blocks_torch(p, device, form), sh.perm, sh.n_ld,
                        specs=sh.block_specs(form, p))
        else:
            eng.load_ld(p, sh.ld_blocks_torch(p, device), sh.perm, sh.n_ld,
                        specs=sh.block_specs())
    drv = SweepDriver()
    drv._setup_driver(eng, Comm(), sh.P, sh.M, 1, sh.chi_local, sh.rank_local, [sh.N_global],
                      log_det, scale_se=scale_se, num_its=100)
    return sh, eng, drv


def _fresh_objective(drv):
    """The objective of the state the driver holds, evaluated again from scratch."""
    eng = drv.engine
    eng.drain()
    eng.eval()
    host = eng.fetch()
    return drv._objective_from(host[eng.layout.totals])


def _ar1_product(sh, p, x):
    """R_p x in SNP order via the O(n) two-sided recursion of an AR(1) matrix."""
    y = np.zeros_like(x)
    for i, blk in enumerate(sh.blocks):
        lo = sh.snp_start[i]
        seg = x[lo:lo + blk.n]
        rho = blk.rho[p]
        fwd = lfilter([1.0], [1.0, -rho], seg)
        bwd = lfilter([1.0], [1.0, -rho], seg[::-1])[::-1]
        y[lo:lo + blk.n] = fwd + bwd - seg
    return y


def test_c3_ld_operator_properties():
    sh, eng, drv = _setup('C3')
    assert sh.N_global > 1_000_000 and len(sh.sizes_all) == 1700
    rng = np.random.default_rng(5)
    x, y = rng.normal(size=(sh.P, sh.N)), rng.normal(size=(sh.P, sh.N))
    rx, ry = eng.ld_matvec(x), eng.ld_matvec(y)
    for p in range(sh.P):                                  # exact closed form, every block
        want = _ar1_product(sh, p, x[p])
        np.testing.assert_allclose(rx[p], want, rtol=1e-10, atol=1e-10)
    assert np.all(rx[:, sh.missing] == 0.0)                # zero rows at LD-missing SNPs
    rz = eng.ld_matvec(2.5 * x - 0.75 * y)                 # linearity
    np.testing.assert_allclose(rz, 2.5 * rx - 0.75 * ry, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose((x * ry).sum(axis=1), (y * rx).sum(axis=1), rtol=1e-10)  # symmetry
    alg, stored = eng.ld_bytes()
    assert alg == sh.ld_bytes and 0.5 * alg < stored < 0.65 * alg     # lower triangle + diag tiles
    eng.close()


def test_c3_fit_invariants():
    sh, eng, drv = _setup('C3')
    drv.initialize_from(sh.fake_mu)       # _initialize's per-SNP part on the device
    state, elbo_prev = None, drv._objective
    assert np.isfinite(elbo_prev)
    for it in range(4):
        state, stats = drv.sweep(state)
        # every accepted update passes the reference's test new >= old - 1e-6|old| - 1e-6, so a
        # sweep (<= 21 accepted updates + M-step) cannot lose more than that many tolerances
        assert state['elbo'] >= elbo_prev - 25 * (1e-6 * abs(elbo_prev) + 1e-6)
        assert np.all(state['L'] >= 1.0) and state['L'][1] == 1.0 and state['L'][2] == 1.0
        assert np.all(np.isfinite(stats)) and stats[0] >= 0
        elbo_prev = state['elbo']
    mean, var = eng.get_moments()
    assert np.all(np.isfinite(mean)) and np.all(var >= 0)
    assert abs(drv._hyper.sum() - 1.0) < 1e-12 and np.all(drv._hyper >= 1e-100)
    # the cached objective equals a fresh evaluation of the same state (no drift in the cache)
    obj = _fresh_objective(drv)
    assert abs(obj - drv._objective) <= 1e-12 * abs(obj)
    eng.close()


def test_c3_learn_scaling_acts_at_full_size_device_decided_equals_host_decided(monkeypatch):
    """--learn-scaling at the headline size, run until it ACTS: from the standard start the full
    C3 problem first gains less than EM_TOL in a sweep after ~158 sweeps (profiles/
    r05c_learn_scaling_where_it_acts.txt); from then on the EVAL decision updates tau
    (_update_error_scaling, reference variational_inference.py:472-486) and re-evaluates.  200
    sweeps decided on the device against 200 decided by the host: with trials that store their
    candidates (VILMA_STASH_LAZY=0) ELBO, L and tau equal to the bit in every sweep; with the default
    lazy trials (the state carried as (stored vi_mu, a, c), written out with the old tau whenever tau
    moves) every L equal, tau and ELBO to rounding; tau moves; every sweep passes the reference's
    acceptance bound; the cached objective equals a fresh evaluation at the final tau."""
    runs = []
    for ahead, stored in ((True, True), (False, True), (True, False)):
        monkeypatch.setenv('VILMA_STASH_LAZY', '0' if stored else '1')
        sh, eng, drv = _setup('C3', scale_se=True)
        drv.initialize_from(sh.fake_mu)
        state, elbo_prev, rows = None, drv._objective, []
        for it in range(200):
            state, stats = drv.sweep(state, lookahead=ahead and it < 199)
            assert state['elbo'] >= elbo_prev - 25 * (1e-6 * abs(elbo_prev) + 1e-6)
            elbo_prev = state['elbo']
            rows.append((state['elbo'], tuple(state['L']), tuple(drv.error_scaling)))
        if ahead:
            assert drv.n_stages_ahead >= 190 and drv.n_stages_skipped == 0
        taus = np.array([r[2] for r in rows])
        assert np.all(taus[:100] == 1.0)                    # nothing to learn while sweeps gain > EM_TOL
        n_moves = int(np.sum(np.any(np.diff(taus, axis=0) != 0.0, axis=1)))
        assert n_moves >= 3 and np.all(np.abs(taus[-1] - 1.0) > 1e-4), (n_moves, taus[-1])
        obj = _fresh_objective(drv)
        assert abs(obj - drv._objective) <= 1e-12 * abs(obj)
        runs.append(rows)
        eng.close()
    assert runs[0] == runs[1]
    for (e, L, tau), (e0, L0, tau0) in zip(runs[2], runs[1]):
        assert L == L0
        assert abs(e - e0) <= 1e-11 * abs(e0)
        np.testing.assert_allclose(tau, tau0, rtol=1e-10)


def test_c3k12_default_mixture_fit_invariants():
    """C3's SNPs and LD under the mixture `vilma fit` builds by default for two cohorts (-K 12 ->
    582 components, reference vi_options.py:17-20, 284-337) at FULL size: 29 GB of vi_mu, the
    per-SNP passes dominate.  The sweeps run from the device's control block although the mixture
    is far beyond the on-chip stash (lazy trials + the materialising sums pass); same invariants as
    C3."""
    sh, eng, drv = _setup('C3K12')
    assert sh.M == 582 and sh.P == 2 and sh.N_global > 1_000_000
    drv.initialize_from(sh.fake_mu)
    state, elbo_prev = None, drv._objective
    assert np.isfinite(elbo_prev)
    for it in range(4):
        state, stats = drv.sweep(state, lookahead=it < 3)
        assert state['elbo'] >= elbo_prev - 25 * (1e-6 * abs(elbo_prev) + 1e-6)
        assert np.all(state['L'] >= 1.0) and state['L'][1] == 1.0
        assert np.all(np.isfinite(stats)) and stats[0] >= 0
        elbo_prev = state['elbo']
    assert drv.n_stages_ahead >= 3 and drv.n_stages_skipped == 0
    mean, var = eng.get_moments()
    assert np.all(np.isfinite(mean)) and np.all(var >= 0)
    assert abs(drv._hyper.sum() - 1.0) < 1e-12 and np.all(drv._hyper >= 1e-100)
    obj = _fresh_objective(drv)
    assert abs(obj - drv._objective) <= 1e-12 * abs(obj)
    eng.close()


def test_c2_three_sweeps_against_oracle():
    from oracle.ldop import EigenBlock, BlockDiagonalLD
    from oracle.vi import MultiPopVIOracle
    from vilma_amd.synthetic import ar1_numpy
    sh, eng, drv = _setup('C2')
    assert sh.N == 100_000 and len(sh.blocks) == 500
    ld = [BlockDiagonalLD([EigenBlock(ar1_numpy(b.n, b.rho[p]), 1.0) for b in sh.blocks],
                          perm=sh.perm, missing=sh.missing) for p in range(sh.P)]
    ovi = MultiPopVIOracle(marginal_effects=sh.betahat, std_errs=sh.se, ld_mats=ld,
                           annotations=np.ones((sh.N, 1)), mixture_covs=list(sh.covs),
                           checkpoint=False, checkpoint_freq=-1, scaled=False, scale_se=False,
                           gwas_N=sh.gwas_N, init_hg=sh.init_hg, num_its=3)
    # closed-form load-time constants == what eigh-based loading derives
    np.testing.assert_allclose(ovi.adj_marginal_effects, sh.adj, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ovi.chi_stat, sh.chi_local, rtol=1e-8)
    np.testing.assert_allclose(ovi.inverse_betas, sh.inverse_betas, rtol=1e-5, atol=1e-9)
    # same start for both (the oracle's own _initialize draws different jitter): the device's
    # _initialize, checked here against the oracle's restatement of it on the same fake_mu
    drv.initialize_from(sh.fake_mu)
    vi_mu0, hyper0 = eng.get_mu(), drv._hyper
    from oracle_engine import OracleEngine
    oeng = OracleEngine(sh.P, sh.N, sh.M, 1)
    oeng.set_snp_data(sh.adj, sh.se, sh.sld, sh.scalings, sh.annot)
    oeng.set_mixture(np.linalg.inv(sh.covs), np.linalg.slogdet(sh.covs)[1])
    osums = oeng.init_state(sh.fake_mu).numpy()
    np.testing.assert_allclose(vi_mu0, oeng.mu, rtol=1e-9, atol=1e-14)
    ohyper = (osums.reshape(1, -1) + 1.) / (osums.sum() + sh.M)
    np.testing.assert_allclose(hyper0, np.maximum(ohyper, 1e-100), rtol=1e-10)
    ovi.nat_grad_vi_delta = None
    from oracle import numerics as nm
    ovi.nat_grad_vi_delta = nm.fast_vi_delta_grad(hyper0, ovi.log_det, ovi.annotations)
    _, d0, _ = ovi._nat_to_not_vi_delta((vi_mu0, None, hyper0))
    oparams = (vi_mu0, d0, hyper0)
    oelbo = ovi.elbo(oparams)
    assert abs(drv._objective - oelbo) < 1e-8 * abs(oelbo)
    state, oL, ored = None, np.ones(5), None
    for it in range(3):
        oparams, oL, oelbo, ored = ovi._optimize_step(oparams, oL, oelbo, 2., ored)
        state, _ = drv.sweep(state)
        assert abs(state['elbo'] - oelbo) < 1e-8 * abs(oelbo), (it, state['elbo'], oelbo)
        assert np.array_equal(state['L'], oL)
    mean, _ = eng.get_moments()
    np.testing.assert_allclose(mean, ovi._posterior_mean(*oparams), rtol=1e-6, atol=1e-10)
    eng.close()


def test_c4_eigen_form_operator_and_fit():
    """BASELINE.json configs[3] at full size with EVERY block in eigen form: ld_eig_fused_kernel
    on multi-slab column-major U (reference LowRankMatrix.dot, matrix_structures.py:148-152) against
    U (s * (U^T x)) formed with torch on a sample of blocks, plus linearity, symmetry, zero rows
    at LD-missing SNPs, and the fit invariants."""
    import torch
    sh, eng, drv = _setup('C4', form='eig')
    assert sh.kind == 'lowrank' and sh.N_global > 1_000_000 and len(sh.sizes_all) == 1700
    alg, stored = eng.ld_bytes()
    assert alg == sh.ld_bytes and alg <= stored < 1.1 * alg            # U once (padded rows) + s
    rng = np.random.default_rng(9)
    x, y = rng.normal(size=(sh.P, sh.N)), rng.normal(size=(sh.P, sh.N))
    rx, ry = eng.ld_matvec(x), eng.ld_matvec(y)
    order = np.argsort(sh.sizes)
    sample = sorted(set(range(0, len(sh.blocks), 40)) | set(order[:3].tolist()) | set(order[-3:].tolist()))
    assert max(int(sh.ranks[i]) for i in sample) > 512                  # several 128-column slabs of U
    for p in range(sh.P):
        for i in sample:
            U, sv = sh._eig[p][i]
            lo = sh.snp_start[i]
            xb = torch.as_tensor(x[p, lo:lo + U.shape[0]], device=U.device)
            want = (U @ (sv * (U.T @ xb))).cpu().numpy()
            np.testing.assert_allclose(rx[p, lo:lo + U.shape[0]], want, rtol=1e-10, atol=1e-10)
    assert np.all(rx[:, sh.missing] == 0.0)
    rz = eng.ld_matvec(2.5 * x - 0.75 * y)
    np.testing.assert_allclose(rz, 2.5 * rx - 0.75 * ry, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose((x * ry).sum(axis=1), (y * rx).sum(axis=1), rtol=1e-10)
    drv.initialize_from(sh.fake_mu)
    state, elbo_prev = None, drv._objective
    assert np.isfinite(elbo_prev)
    for it in range(3):
        state, stats = drv.sweep(state)
        assert state['elbo'] >= elbo_prev - 25 * (1e-6 * abs(elbo_prev) + 1e-6)
        assert np.all(np.isfinite(stats))
        elbo_prev = state['elbo']
    obj = _fresh_objective(drv)
    assert abs(obj - drv._objective) <= 1e-12 * abs(obj)
    eng.close()


def test_c4f_factor_model_recipe_at_full_size():
    """SURVEY.md 8(d)'s own recipe for configs[3] at FULL size: per block and cohort
    R = D^-1/2 (F F^T / m + 0.05 I) D^-1/2, m = ceil(n / 4), eigendecomposed and cut by the loader's
    --ldthresh 0.8 rule -- ranks are data-derived (about n / 4), per cohort.  `auto` storage: the
    operator against U (s * (U^T x)) on a sample of blocks, symmetry, linearity, zero rows at the
    LD-missing SNPs, and the fit invariants over three sweeps."""
    import torch
    sh, eng, drv = _setup('C4f', form='auto')
    assert sh.kind == 'lowrank' and sh.spectrum == 'factor'
    assert sh.N_global > 1_000_000 and len(sh.sizes_all) == 1700
    frac = sum(float(r.sum()) for r in sh.ranks_by_cohort) / (sh.P * float(sh.sizes.sum()))
    assert 0.2 < frac < 0.35                                            # kept rank ~ n / 4
    rng = np.random.default_rng(10)
    x, y = rng.normal(size=(sh.P, sh.N)), rng.normal(size=(sh.P, sh.N))
    rx, ry = eng.ld_matvec(x), eng.ld_matvec(y)
    order = np.argsort(sh.sizes)
    sample = sorted(set(range(0, len(sh.blocks), 50)) | set(order[:3].tolist()) | set(order[-3:].tolist()))
    for p in range(sh.P):
        for i in sample:
            U, sv = sh._eig[p][i]
            lo = sh.snp_start[i]
            xb = torch.as_tensor(x[p, lo:lo + U.shape[0]], device=U.device)
            want = (U @ (sv * (U.T @ xb))).cpu().numpy()
            np.testing.assert_allclose(rx[p, lo:lo + U.shape[0]], want, rtol=1e-10, atol=1e-10)
    assert np.all(rx[:, sh.missing] == 0.0)
    rz = eng.ld_matvec(2.5 * x - 0.75 * y)
    np.testing.assert_allclose(rz, 2.5 * rx - 0.75 * ry, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose((x * ry).sum(axis=1), (y * rx).sum(axis=1), rtol=1e-10)
    drv.initialize_from(sh.fake_mu)
    state, elbo_prev = None, drv._objective
    assert np.isfinite(elbo_prev)
    for it in range(3):
        state, stats = drv.sweep(state, lookahead=it < 2)
        assert state['elbo'] >= elbo_prev - 25 * (1e-6 * abs(elbo_prev) + 1e-6)
        assert np.all(state['L'] >= 1.0) and np.all(np.isfinite(stats))
        elbo_prev = state['elbo']
    obj = _fresh_objective(drv)
    assert abs(obj - drv._objective) <= 1e-12 * abs(obj)
    eng.close()


def test_c5_operator_and_fit_invariants():
    """BASELINE.json configs[4] on ONE GPU: P=4 (Cholesky branch of the per-SNP pass), 34 000
    block-cohorts, ~100 GB resident.  Same size-independent properties as C3."""
    sh, eng, drv = _setup('C5')
    assert sh.P == 4 and sh.M == 81 and sh.N_global > 5_000_000 and len(sh.sizes_all) == 8500
    rng = np.random.default_rng(6)
    x = rng.normal(size=(sh.P, sh.N))
    rx = eng.ld_matvec(x)
    for p in range(sh.P):
        np.testing.assert_allclose(rx[p], _ar1_product(sh, p, x[p]), rtol=1e-10, atol=1e-10)
    assert np.all(rx[:, sh.missing] == 0.0)
    drv.initialize_from(sh.fake_mu)
    state, elbo_prev = None, drv._objective
    assert np.isfinite(elbo_prev)
    for it in range(3):
        state, stats = drv.sweep(state)
        assert state['elbo'] >= elbo_prev - 25 * (1e-6 * abs(elbo_prev) + 1e-6)
        assert np.all(np.isfinite(stats))
        elbo_prev = state['elbo']
    mean, var = eng.get_moments()
    assert np.all(np.isfinite(mean)) and np.all(var >= 0)
    obj = _fresh_objective(drv)
    assert abs(obj - drv._objective) <= 1e-12 * abs(obj)
    eng.close()


def _sample_against_oracle(workload, block_range, n_sweeps, form='auto', scale_se=False, warm_sweeps=0):
    """A contiguous run of a full-size workload's LD blocks (its SNPs, LD, sumstats and mixture
    exactly as in the full problem) fitted on the GPU -- sweeps queued ahead and decided on the
    device -- and by the oracle from the same starting point: ELBO 1e-9 per sweep, the L
    trajectory to the bit, posterior means 1e-7 (north star: 1e-5).  warm_sweeps: the GPU first
    runs that many sweeps alone and the comparison starts from the state it reached (its vi_mu,
    hyper_delta, L, running ELBO change handed to the oracle) -- how --learn-scaling is reached
    where it acts: tau only moves once a sweep gains less than EM_TOL.  Returns the error scaling
    after every compared sweep."""
    import os
    from oracle.ldop import EigenBlock, BlockDiagonalLD
    from oracle.vi import MultiPopVIOracle
    from oracle import native, numerics as nm
    from vilma_amd.synthetic import ar1_numpy
    sh, eng, drv = _setup(workload, form=form, block_range=block_range, scale_se=scale_se)
    assert len(sh.blocks) == block_range[1] - block_range[0]
    assert len(sh.blocks) >= 0.1 * len(sh.sizes_all)
    if sh.kind == 'lowrank' and sh.spectrum == 'factor':
        # The oracle decomposes the sample's DENSE blocks itself: each block's matrix is formed again
        # on the host from its seeded stream (SURVEY 8d's recipe, numpy) and goes through the
        # oracle's own restatement of _svd_threshold / LowRankMatrix.__init__ with host LAPACK
        # (matrix_structures.py:15-28, 95-146).  The product's factors come from its GPU loader
        # (rocSOLVER stacks + select_eigenpairs): the comparison below pins the decomposition and the
        # kept ranks as well as the sweep.
        from concurrent.futures import ThreadPoolExecutor
        from threadpoolctl import threadpool_limits
        from vilma_amd.synthetic import FACTOR_LD_THRESH

        def host_block(ip):
            i, p = ip
            n = sh.blocks[i].n
            m = -(-n // 4)
            F = np.random.default_rng([sh.seed, 5000 + sh.b0 + i, p]).normal(size=(n, m))
            R = F @ F.T / m
            R[np.diag_indices(n)] += 0.05
            d = 1.0 / np.sqrt(np.diag(R))
            R = d[:, None] * R * d[None, :]
            return EigenBlock(X=0.5 * (R + R.T), t=FACTOR_LD_THRESH)
        with threadpool_limits(limits=1), ThreadPoolExecutor(max_workers=16) as pool:
            host = list(pool.map(host_block, [(i, p) for p in range(sh.P) for i in range(len(sh.blocks))]))
        nb = len(sh.blocks)
        for p in range(sh.P):
            for i in range(nb):
                assert host[p * nb + i].s.size == sh._eig[p][i][0].shape[1], (p, i)    # kept ranks agree
        ld = [BlockDiagonalLD(host[p * nb:(p + 1) * nb], perm=sh.perm, missing=sh.missing)
              for p in range(sh.P)]
    elif sh.kind == 'lowrank':
        ld = [BlockDiagonalLD([EigenBlock(u=U.cpu().numpy(), s=sv.cpu().numpy(), t=1.0)
                               for U, sv in sh._eig[p]], perm=sh.perm, missing=sh.missing)
              for p in range(sh.P)]
    else:
        ld = [BlockDiagonalLD([EigenBlock(ar1_numpy(b.n, b.rho[p]), 1.0) for b in sh.blocks],
                              perm=sh.perm, missing=sh.missing) for p in range(sh.P)]
    ovi = MultiPopVIOracle(marginal_effects=sh.betahat, std_errs=sh.se, ld_mats=ld,
                           annotations=np.ones((sh.N, 1)), mixture_covs=list(sh.covs),
                           checkpoint=False, checkpoint_freq=-1, scaled=False, scale_se=scale_se,
                           gwas_N=sh.gwas_N, init_hg=sh.init_hg, num_its=n_sweeps)
    # (the sample's constants use its own sum of 1/se^2, on both sides)
    np.testing.assert_allclose(ovi.adj_marginal_effects, sh.adj, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ovi.chi_stat, sh.chi_local, rtol=1e-8)
    # one starting point for both: the device's _initialize on the shard's own jittered start
    drv.initialize_from(sh.fake_mu)
    state = None
    for it in range(warm_sweeps):
        state, _ = drv.sweep(state, lookahead=True)
    vi_mu0, hyper0 = eng.get_mu(), drv._hyper       # (reading the state takes back what was queued ahead)
    if warm_sweeps:
        assert np.all(drv.error_scaling == 1.0), 'tau moved during the warm-up: start the comparison earlier'
    ovi.nat_grad_vi_delta = nm.fast_vi_delta_grad(hyper0, ovi.log_det, ovi.annotations)
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    native.enable(threads=threads)          # the oracle's per-SNP passes as compiled C loops
    taus = []
    try:
        _, d0, _ = ovi._nat_to_not_vi_delta((vi_mu0, None, hyper0))
        oparams = (vi_mu0, d0, hyper0)
        oelbo = ovi.elbo(oparams)
        assert abs(drv._objective - oelbo) < 1e-9 * abs(oelbo)
        oL, ored = np.ones(5), None
        if state is not None:
            # the oracle continues the GPU's run: same L, same running change; its ELBO carries on
            # from the GPU's accumulated value, as optimize() would carry it
            oL, ored, oelbo = state['L'].copy(), state['running'], state['elbo']
        ahead = 0
        for it in range(n_sweeps):
            oparams, oL, oelbo, ored = ovi._optimize_step(oparams, oL, oelbo, 2., ored)
            a0 = drv.n_stages_ahead
            state, _ = drv.sweep(state, lookahead=it + 1 < n_sweeps)
            ahead += drv.n_stages_ahead - a0
            assert abs(state['elbo'] - oelbo) < 1e-9 * abs(oelbo), (it, state['elbo'], oelbo)
            assert np.array_equal(state['L'], oL), (it, state['L'], oL)
            assert abs(state['running'] - ored) <= 1e-9 * abs(ored)
            if scale_se:
                np.testing.assert_allclose(drv.error_scaling, ovi.error_scaling, rtol=1e-9)
            taus.append(np.array(drv.error_scaling))
        assert ahead >= n_sweeps - 2            # the sweeps did run from the device's control block
        mean, var = eng.get_moments()
        omean = ovi._posterior_mean(*oparams)
        np.testing.assert_allclose(mean, omean, rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(var, ovi._posterior_marginal_variance(omean, *oparams[:2]),
                                   rtol=1e-7, atol=1e-16)
        np.testing.assert_allclose(drv._hyper, oparams[2], rtol=1e-8, atol=1e-300)
    finally:
        native.disable()
        eng.close()
    return taus


def test_c3_block_sample_four_sweeps_against_the_oracle():
    """170 consecutive blocks of C3's 1700 (10 %, ~100 k SNPs x 2 cohorts, M = 40), four sweeps."""
    _sample_against_oracle('C3', (700, 870), 4)


def test_c3_block_sample_learn_scaling_where_tau_moves_against_the_oracle():
    """--learn-scaling where it acts (reference variational_inference.py:441-448, 472-486, 712-738):
    the error scaling only moves in a sweep that gained less than EM_TOL = 10, which this sample of
    C3 reaches after ~50 sweeps (profiles/r05c_learn_scaling_where_it_acts.txt; the full problem
    after ~158).  The GPU runs 46 sweeps with scale_se on its own, hands its state to the oracle,
    and both run 8 more: in those the EVAL decision of the device's stage machine takes the tau
    update and the re-evaluation behind it -- tau 1e-9, ELBO 1e-9, L to the bit, every sweep."""
    taus = _sample_against_oracle('C3', (700, 870), 8, scale_se=True, warm_sweeps=46)
    moved = [t for t in taus if np.any(t != 1.0)]
    assert len(moved) >= 2, taus                 # tau did move, in more than one sweep
    assert not np.array_equal(moved[0], moved[-1])


def test_c4f_block_sample_three_sweeps_against_the_oracle():
    """170 consecutive blocks of C4f (SURVEY 8d's factor-model LD cut at --ldthresh 0.8; eigen-form
    and dense blocks mixed by the `auto` rule), three sweeps."""
    _sample_against_oracle('C4f', (700, 870), 3)
