"""GPU parity, kernel level: every C-ABI evaluation entry point against the oracle on the
golden problems (same seeded inputs).  Tolerances: 1e-9 relative on sums (the north-star bar is
1e-5 on ELBO/posterior means; fp64 kernels with a different summation order land ~1e-13)."""
import numpy as np
import pytest
import torch

from helpers import (golden, oracle_from_traj, engine_from_oracle, oracle_totals, TRAJ_NAMES)

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-9, atol=1e-9):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


@pytest.mark.parametrize('form', ['dense', 'eig', 'auto'])
@pytest.mark.parametrize('name', TRAJ_NAMES)
def test_ld_matvec(name, form):
    """vilma_ld_matvec == BlockDiagonalMatrix.dot incl. perm/missing (matrix_structures.py:389-408)."""
    g = golden('traj_%s.npz' % name)
    vi, ld = oracle_from_traj(g)
    eng = engine_from_oracle(vi, ld, form)
    rng = np.random.default_rng(3)
    x = rng.normal(size=(vi.num_pops, vi.num_loci))
    want = np.stack([ld[p].dot(x[p]) for p in range(vi.num_pops)])
    got = eng.ld_matvec(x)
    _close(got, want, rtol=1e-11, atol=1e-11)
    assert np.all(got[:, g['missing']] == 0.0)
    for p in range(vi.num_pops):           # one cohort at a time leaves the others untouched
        solo = eng.ld_matvec(x, cohort=p)
        _close(solo[p], want[p], rtol=1e-11, atol=1e-11)
    alg, stored = eng.ld_bytes()
    # dense symmetric blocks keep the lower triangle (n^2/2 + 64 n elements), eigen-form blocks
    # U and diag(s)U^T (2 n r), both with rows padded to 128 B
    assert alg > 0 and stored > 0.4 * alg
    eng.close()


@pytest.mark.parametrize('form', ['dense', 'eig'])
@pytest.mark.parametrize('name', TRAJ_NAMES)
def test_eval_trial_and_sums(name, form):
    from oracle import numerics as nm
    g = golden('traj_%s.npz' % name)
    vi, ld = oracle_from_traj(g)
    eng = engine_from_oracle(vi, ld, form)
    np.random.seed(int(g['seed']))
    vi_mu, vi_delta, hyper = vi._initialize()
    eng.set_hyper(hyper)
    eng.set_mu(vi_mu)

    # ---- vilma_eval at the initial point
    tot = eng.eval().cpu().numpy()
    want = oracle_totals(vi, (vi_mu, vi_delta, hyper))
    _close(tot, want, rtol=1e-9, atol=1e-8)
    eng.accept(False)
    mean, var = eng.get_moments()
    omean = vi._posterior_mean(vi_mu, vi_delta)
    _close(mean, omean, rtol=1e-10, atol=1e-14)
    _close(var, vi._posterior_marginal_variance(omean, vi_mu, vi_delta), rtol=1e-9, atol=1e-16)
    _close(eng.get_delta(), vi_delta, rtol=1e-9, atol=1e-300)
    _close(eng.delta_sums().cpu().numpy().reshape(vi.num_annotations, vi.num_mix),
           nm.sum_annotations(vi_delta, vi.annotations, vi.num_annotations), rtol=1e-10)

    # ---- one line-search trial of _update_beta at step 1/L, L = 1 and 2
    old_nat = nm.fast_nat_inner_product_m2(vi_mu, vi.nat_sigma)
    grad = vi._nat_grad_beta(vi_mu, vi_delta, hyper)
    const = np.copy(vi.vi_sigma_log_det.T)
    for step in (1.0, 0.5):
        nat_mu = nm.sum_betas(old_nat, grad, step)
        new_mu = nm.fast_nat_inner_product(nat_mu, vi.vi_sigma)
        new_delta = nm.fast_invert_nat_vi_delta(new_mu, nat_mu, const, vi.nat_grad_vi_delta)
        tot = eng.trial(step).cpu().numpy()
        _close(tot, oracle_totals(vi, (new_mu, new_delta, hyper)), rtol=1e-9, atol=1e-8)
        # the M-step statistic of the candidate, two ways: the per-tile sums its own per-SNP pass
        # left behind (no second pass over vi_mu) and delta_kernel
        want_sums = nm.sum_annotations(new_delta, vi.annotations, vi.num_annotations)
        if eng.trial_sums_available():
            _close(eng.trial_sums().cpu().numpy().reshape(want_sums.shape), want_sums, rtol=1e-10)
        _close(eng.delta_sums(1).cpu().numpy().reshape(want_sums.shape), want_sums, rtol=1e-10)
    eng.accept(True)
    _close(eng.get_mu(), new_mu, rtol=1e-9, atol=1e-16)
    _close(eng.get_delta(), new_delta, rtol=1e-8, atol=1e-300)
    _close(eng.get_moments()[0], vi._posterior_mean(new_mu, new_delta), rtol=1e-9, atol=1e-14)

    # ---- hyper update: new table, re-evaluate without moving vi_mu
    new_hyper = nm.sum_annotations(new_delta, vi.annotations, vi.num_annotations)
    new_hyper = np.maximum(new_hyper / (vi.annotation_counts.reshape((-1, 1)) + 1e-100), 1e-100)
    new_hyper /= new_hyper.sum(axis=1, keepdims=True)
    vi.nat_grad_vi_delta = nm.fast_vi_delta_grad(new_hyper, vi.log_det, vi.annotations)
    _, d2, _ = vi._nat_to_not_vi_delta((new_mu, new_delta, new_hyper))
    eng.set_hyper(new_hyper)
    tot = eng.eval().cpu().numpy()
    _close(tot, oracle_totals(vi, (new_mu, d2, new_hyper)), rtol=1e-9, atol=1e-8)
    eng.accept(False)

    # ---- error-scaling change re-derives every sigma-dependent constant on the fly
    vi.error_scaling = np.linspace(0.8, 1.3, vi.num_pops)
    vi._set_vi_sigma()
    _, d3, _ = vi._nat_to_not_vi_delta((new_mu, d2, new_hyper))
    eng.set_tau(vi.error_scaling)
    tot = eng.eval().cpu().numpy()
    _close(tot, oracle_totals(vi, (new_mu, d3, new_hyper)), rtol=1e-9, atol=1e-8)
    eng.close()


def test_mean_diff():
    g = golden('traj_p2_scale_se.npz')
    vi, ld = oracle_from_traj(g)
    eng = engine_from_oracle(vi, ld)
    np.random.seed(1)
    vi_mu, vi_delta, hyper = vi._initialize()
    eng.set_hyper(hyper)
    eng.set_mu(vi_mu)
    eng.eval(); eng.accept(False)
    old = vi._posterior_mean(vi_mu, vi_delta) * vi.scalings
    eng.snapshot_mean()
    eng.trial(0.7); eng.accept(True)
    new = eng.get_moments()[0] * vi.scalings
    d = torch.cat(eng.mean_diff()).cpu().numpy()
    diff = np.abs(new - old)
    assert d[0] == np.sum(diff > 1e-6 + 1e-6 * np.abs(old))
    _close(d[1], diff.sum(), rtol=1e-10)
    _close(d[2], (diff ** 2).sum(), rtol=1e-10)
    _close(d[3], np.abs(new).max(), rtol=1e-12)
    _close(d[4], diff.max(), rtol=1e-12)
    _close(d[5], np.abs((new - old) / (old + 1e-100)).max(), rtol=1e-9)
    d2 = torch.cat(eng.mean_diff()).cpu().numpy()        # snapshot was replaced: no change now
    assert d2[0] == 0 and d2[4] == 0
    eng.close()


@pytest.mark.parametrize('name', ['p2_scale_se', 'p4_m81', 'p1_scaled'])
def test_eval_with_fused_convergence_statistics(name):
    """vilma_eval_diff == vilma_eval followed by vilma_mean_diff, bit for bit in the sums it
    shares and to rounding in the statistics (different reduction tree), and it moves the
    snapshot the same way."""
    g = golden('traj_%s.npz' % name)
    vi, ld = oracle_from_traj(g)
    np.random.seed(2)
    vi_mu, vi_delta, hyper = vi._initialize()
    outs = []
    for fused in (False, True):
        eng = engine_from_oracle(vi, ld)
        eng.set_hyper(hyper)
        eng.set_mu(vi_mu)
        eng.eval(); eng.accept(False)
        eng.snapshot_mean()
        eng.trial(0.6); eng.accept(True)
        eng.set_hyper(np.roll(hyper, 1, axis=1))            # an "M-step": new hyper, same vi_mu
        if fused:
            tot = eng.eval(diff=True).cpu().numpy().copy(); eng.accept(False)
            d = np.concatenate([eng._dsum.cpu().numpy(), eng._dmax.cpu().numpy()])
        else:
            tot = eng.eval().cpu().numpy().copy(); eng.accept(False)
            d = torch.cat(eng.mean_diff()).cpu().numpy()
        d_again = torch.cat(eng.mean_diff()).cpu().numpy()  # the snapshot is now this state
        outs.append((tot, d, d_again, eng.get_moments()[0]))
        eng.close()
    (t0, d0, a0, m0), (t1, d1, a1, m1) = outs
    assert np.array_equal(t0, t1) and np.array_equal(m0, m1)
    assert d0[0] == d1[0] and np.array_equal(d0[3:], d1[3:])
    _close(d1[1:3], d0[1:3], rtol=1e-12)
    assert a0[0] == 0 and a1[0] == 0 and a0[4] == 0 and a1[4] == 0


@pytest.mark.parametrize('form', ['dense', 'eig'])
@pytest.mark.parametrize('name', ['p2_bigblock', 'p2_bigblock_lr', 'p4_m81', 'p1_scaled', 'p2_scale_se'])
def test_two_step_trial_equals_two_trials(name, form):
    """vilma_trial_beta2 evaluates two step sizes in one pass over vi_mu and the LD store; each
    candidate's sums, vi_mu and moments are bit-identical to a one-step trial at that step, and
    either can be accepted (take_mu = 1 / 2)."""
    g = golden('traj_%s.npz' % name)
    vi, ld = oracle_from_traj(g)
    np.random.seed(5)
    vi_mu, vi_delta, hyper = vi._initialize()

    def fresh():
        eng = engine_from_oracle(vi, ld, form)
        eng.set_hyper(hyper)
        eng.set_mu(vi_mu)
        eng.eval(); eng.accept(False)
        return eng
    sa, sb = 0.8, 0.4
    single = {}
    for step in (sa, sb):
        eng = fresh()
        tot = eng.trial(step).cpu().numpy().copy()
        tile = eng.trial_sums().cpu().numpy().copy() if eng.trial_sums_available() else None
        sums = eng.delta_sums(1).cpu().numpy().copy()
        eng.accept(True)
        single[step] = (tot, sums, eng.get_mu(), eng.get_moments(), eng.get_delta(), tile)
        eng.close()
    for take in (1, 2):
        eng = fresh()
        ta, tb = eng.trial2(sa, sb)
        ta, tb = ta.cpu().numpy().copy(), tb.cpu().numpy().copy()
        assert np.array_equal(ta, single[sa][0]) and np.array_equal(tb, single[sb][0])
        if eng.trial_sums_available() == 2:
            # both candidates' responsibility sums out of the one pass: bit-identical to the
            # one-step trial's, and the same numbers delta_kernel derives from the stored vi_mu
            tile_a, tile_b = (t.cpu().numpy().copy() for t in eng.trial_sums(both=True))
            assert np.array_equal(tile_a, single[sa][5]) and np.array_equal(tile_b, single[sb][5])
            np.testing.assert_allclose(tile_a, single[sa][1], rtol=1e-12)
            np.testing.assert_allclose(tile_b, single[sb][1], rtol=1e-12)
        sums = eng.delta_sums(1 if take == 1 else 3).cpu().numpy().copy()
        eng.accept(take)
        want = single[sa if take == 1 else sb]
        assert np.array_equal(sums, want[1])
        assert np.array_equal(eng.get_mu(), want[2])
        m, v = eng.get_moments()
        assert np.array_equal(m, want[3][0]) and np.array_equal(v, want[3][1])
        assert np.array_equal(eng.get_delta(), want[4])
        # the accepted candidate is a full state: the next trial starts from it
        nxt = eng.trial(0.5).cpu().numpy().copy()
        eng2 = fresh()
        eng2.trial(sa if take == 1 else sb); eng2.accept(True)
        assert np.array_equal(nxt, eng2.trial(0.5).cpu().numpy())
        eng.close(); eng2.close()
    eng = fresh()
    eng.trial(sa)
    with pytest.raises(Exception):
        eng.accept(2)                       # no second candidate after a one-step trial
    eng.close()


def test_errors_are_loud():
    from vilma_amd import _lib
    from vilma_amd.engine import HipEngine
    with pytest.raises(_lib.VilmaHipError):
        HipEngine(9, 10, 3, 1)              # more than 8 cohorts: unsupported
    eng = HipEngine(5, 8, 3, 1)             # five cohorts: one candidate per trial only
    eng.set_snp_data(np.ones((5, 8)), np.ones((5, 8)), np.ones((5, 8)), np.ones((5, 8)),
                     np.zeros(8, dtype=np.int32))
    eng.set_mixture(np.stack([np.eye(5), 2 * np.eye(5), 3 * np.eye(5)]), np.zeros(3))
    eng.set_hyper(np.full((1, 3), 1 / 3))
    for p in range(5):
        eng.load_ld(p, [('dense', np.eye(8))], np.arange(8, dtype=np.int64), 8)
    eng.set_mu(np.zeros((3, 5, 8)))
    eng.eval(); eng.accept(False)
    eng.trial(0.5)
    with pytest.raises(_lib.VilmaHipError, match='up to four cohorts'):
        eng.trial2(0.5, 0.25)
    eng.close()
    eng = HipEngine(1, 10, 3, 1)
    with pytest.raises(_lib.VilmaHipError):
        eng.eval()                          # LD not loaded
    with pytest.raises(_lib.VilmaHipError):
        eng.load_ld(0, [('dense', np.eye(4))], np.array([0, 1, 2, 3, 3, 5, 6, 7, 8, 9]), 4)
    eng.close()


@pytest.mark.parametrize('rank_of', [lambda n: 1, lambda n: min(n, 17), lambda n: max(1, n // 4),
                                     lambda n: n])
def test_fused_eigen_product_ranks_and_two_right_hand_sides(rank_of):
    """Eigen-form blocks of every panel-width class with ranks below one panel, ragged, a quarter
    of the block and full: the operator against numpy, and a two-step trial (two right-hand sides
    in one pass over U) bit-identical to two one-step trials."""
    from vilma_amd.engine import HipEngine
    sizes = [37, 400, 530, 1100, 2100, 3500]      # (3 500: the tall class, 512-thread workgroups)
    rng = np.random.default_rng(7)
    n_ld = sum(sizes)
    N, P, M = n_ld + 11, 2, 3
    perm = rng.permutation(N).astype(np.int64)
    eigs = []
    for p in range(P):
        row = []
        for n in sizes:
            r = rank_of(n)
            U = np.linalg.qr(rng.normal(size=(n, r)))[0]
            row.append((U, rng.uniform(0.1, 2.0, size=r)))
        eigs.append(row)
    eng = HipEngine(P, N, M, 1)
    for p in range(P):
        eng.load_ld(p, [('eig', U, s) for U, s in eigs[p]], perm, n_ld)
    x = rng.normal(size=(P, N))
    got = eng.ld_matvec(x)
    want = np.zeros((P, N))
    for p in range(P):
        lo = 0
        for n, (U, s) in zip(sizes, eigs[p]):
            idx = perm[lo:lo + n]
            want[p, idx] = U @ (s * (U.T @ x[p, idx]))
            lo += n
    _close(got, want, rtol=1e-11, atol=1e-11)
    # a state to run trials from
    se = rng.uniform(0.005, 0.02, size=(P, N))
    ldd = np.zeros((P, N)); ldd[:, perm[:n_ld]] = 1.0
    adj = rng.normal(size=(P, N)) / se
    adj[:, perm[n_ld:]] = 0.0
    eng.set_snp_data(adj, se, ldd / se ** 2, np.ones((P, N)), np.zeros(N, dtype=np.int32))
    covs = np.stack([v * np.eye(P) for v in (1e-6, 1e-4, 1e-2)])
    eng.set_mixture(np.linalg.inv(covs), np.linalg.slogdet(covs)[1])
    eng.set_hyper(np.full((1, M), 1.0 / M))
    mu = rng.normal(size=(M, P, N)) * 1e-3
    mu[:, :, perm[n_ld:]] = 0.0
    eng.set_mu(mu)
    eng.eval(); eng.accept(False)
    one = {}
    for step in (0.7, 0.35):
        one[step] = eng.trial(step).cpu().numpy().copy()
    ta, tb = eng.trial2(0.7, 0.35)
    assert np.array_equal(ta.cpu().numpy(), one[0.7])
    assert np.array_equal(tb.cpu().numpy(), one[0.35])
    eng.close()


@pytest.mark.parametrize('sizes', [[1, 2, 3], [127, 128, 129], [255, 256, 257], [300, 64, 700],
                                   [1000, 17], [511, 513, 512], [2431],
                                   # the fused eigen-form product holds 2, 4, 8 or 12 rows per
                                   # thread (blocks up to 512, 1024, 2048, 3072 rows; up to 6 144
                                   # with workgroups of 512 threads); taller blocks keep the
                                   # two-pass kernels
                                   [512, 513, 9], [1024, 1025], [2048, 2049], [3072, 5], [3073, 70],
                                   [6144, 3], [6145, 4100]])
def test_ld_matvec_block_sizes(sizes):
    """The symmetric (lower-triangle) dense kernel and the eigen-form kernels across slab
    boundaries: block sizes around multiples of 128, odd sizes, ragged mixes, perm + missing."""
    from vilma_amd.engine import HipEngine
    rng = np.random.default_rng(sum(sizes))
    n_ld = sum(sizes)
    N = n_ld + 7
    perm = rng.permutation(N).astype(np.int64)
    mats, eigs = [], []
    for n in sizes:
        r = max(1, n // 3)
        U = np.linalg.qr(rng.normal(size=(n, r)))[0]
        s = rng.uniform(0.1, 2.0, size=r)
        mats.append((U * s) @ U.T + np.diag(rng.uniform(0.1, 1.0, size=n)))   # full-rank SPD
        eigs.append((U, s))
    x = rng.normal(size=(2, N))
    eng = HipEngine(2, N, 3, 1)
    eng.load_ld(0, [('dense', R) for R in mats], perm, n_ld)
    eng.load_ld(1, [('eig', U, s) for U, s in eigs], perm, n_ld)
    got = eng.ld_matvec(x)
    want = np.zeros((2, N))
    lo = 0
    for n, R, (U, s) in zip(sizes, mats, eigs):
        idx = perm[lo:lo + n]
        want[0, idx] = R @ x[0, idx]
        want[1, idx] = U @ (s * (U.T @ x[1, idx]))
        lo += n
    _close(got, want, rtol=1e-11, atol=1e-11)
    assert np.all(got[:, perm[n_ld:]] == 0.0)
    # symmetry property of the operator: x^T (R y) == y^T (R x)
    y = rng.normal(size=(2, N))
    ry = eng.ld_matvec(y)
    _close((x * ry).sum(axis=1), (y * got).sum(axis=1), rtol=1e-10)
    alg, stored = eng.ld_bytes()
    assert alg == 8 * sum(n * n + n * max(1, n // 3) for n in sizes)
    # the measurement yardstick (vilma_prof_stream_store): a bare read of the stores, whole
    # 32 KB steps of each cohort's store, and the product still gives the same answer afterwards
    ms, nbytes = eng.stream_store(2)
    assert 0 <= nbytes <= stored and nbytes % 32768 == 0 and ms >= 0 and (ms > 0 or nbytes == 0)
    assert np.array_equal(eng.ld_matvec(x), got)
    eng.close()


@pytest.mark.parametrize('tile', ['0', '512,1', '512,2', '512,4', '256,2', '128,1', 'auto'])
def test_ld_tile_shapes_and_two_right_hand_sides(tile, monkeypatch):
    """The tiled symmetric product (ld_tile_kernel) in every strip shape, the decomposition of round 4
    (VILMA_LD_TILE=0) and the automatic choice: against numpy, and two right-hand sides in one pass
    (vilma_ld_matvec2) bit-identical to two passes -- launch after launch (round 5: a partial-sum
    store of the first right-hand side had its data registers overwritten by the second one's
    arithmetic, a few entries per launch, in shapes with rows below the diagonal tile)."""
    from vilma_amd.engine import HipEngine
    if tile == 'auto':
        monkeypatch.delenv('VILMA_LD_TILE', raising=False)
    else:
        monkeypatch.setenv('VILMA_LD_TILE', tile)
    rng = np.random.default_rng(11)
    sizes = [200] * 60 + [700, 300, 1025, 64, 513, 129, 2431, 512, 1, 127]
    n_ld = sum(sizes)
    N = n_ld + 5
    perm = rng.permutation(N).astype(np.int64)
    mats = []
    for n in sizes:
        A = rng.normal(size=(n, n)) / np.sqrt(n)
        mats.append(A @ A.T + np.eye(n))
    eng = HipEngine(1, N, 3, 1)
    eng.load_ld(0, [('dense', R) for R in mats], perm, n_ld)
    rows, slabs, items = eng.ld_tile()
    if tile == 'auto':
        assert (rows, slabs) == (512, 1)            # 70 blocks: far fewer items than workgroup slots
    elif tile == '0':
        assert (rows, slabs) == (0, 0)
    else:
        assert (rows, slabs) == tuple(int(v) for v in tile.split(','))
    assert items >= len(sizes)
    for rep in range(6):
        xa, xb = rng.normal(size=(1, N)), rng.normal(size=(1, N))
        ya, yb = eng.ld_matvec(xa), eng.ld_matvec(xb)
        want = np.zeros((2, N))
        lo = 0
        for n, R in zip(sizes, mats):
            idx = perm[lo:lo + n]
            want[0, idx] = R @ xa[0, idx]
            want[1, idx] = R @ xb[0, idx]
            lo += n
        _close(ya[0], want[0], rtol=1e-11, atol=1e-11)
        _close(yb[0], want[1], rtol=1e-11, atol=1e-11)
        y2a, y2b = eng.ld_matvec2(xa, xb)
        assert np.array_equal(y2a, ya) and np.array_equal(y2b, yb)
    eng.close()


def test_device_mstep_matches_host_formula():
    """vilma_mstep == the M-step of _update_hyper_delta (variational_inference.py:832-848)."""
    from oracle import numerics as nm
    g = golden('traj_p2_scale_se.npz')            # A = 2 annotations
    vi, ld = oracle_from_traj(g)
    eng = engine_from_oracle(vi, ld)
    eng.set_annotation_counts(vi.annotation_counts)
    np.random.seed(int(g['seed']))
    vi_mu, vi_delta, hyper = vi._initialize()
    eng.set_hyper(hyper)
    eng.set_mu(vi_mu)
    eng.eval(); eng.accept(False)
    sums = eng.delta_sums()
    hyper_dev = eng.mstep(sums).cpu().numpy().reshape(vi.num_annotations, vi.num_mix)
    want = nm.sum_annotations(vi_delta, vi.annotations, vi.num_annotations)
    want = np.maximum(want / (vi.annotation_counts.reshape((-1, 1)) + 1e-100), 1e-100)
    want /= want.sum(axis=1, keepdims=True)
    _close(hyper_dev, want, rtol=1e-12, atol=1e-300)
    # the table installed by mstep is the one set_hyper(want) would install
    t1 = eng.eval().cpu().numpy().copy()
    eng.set_hyper(want)
    t2 = eng.eval().cpu().numpy().copy()
    _close(t1, t2, rtol=1e-12, atol=1e-10)
    eng.close()
