"""GPU parity on the edges of the problem space: minimal sizes, one-SNP blocks, a cohort without
any LD, an annotation nobody has, many mixture components, three cohorts, big blocks.  Each case
runs the product class API on the HIP engine against the oracle for a few sweeps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ar1(n, rho):
    i = np.arange(n)
    return rho ** np.abs(i[:, None] - i[None, :])


def _problem(rng, P, sizes_per_cohort, N, M, A=1, ldthresh=1.0, empty_annot=False):
    """Random SPD blocks per cohort on random disjoint SNP subsets (different partitions per
    cohort), sumstats from the model's own likelihood shape."""
    perms, missings, blocks = [], [], []
    for p in range(P):
        sizes = sizes_per_cohort[p]
        n_ld = int(np.sum(sizes))
        order = rng.permutation(N)
        perms.append(np.concatenate([order[:n_ld], np.sort(order[n_ld:])]).astype(np.int64))
        missings.append(np.sort(order[n_ld:]).astype(np.int64))
        blocks.append([_ar1(n, rng.uniform(0.2, 0.9)) for n in sizes])
    se = rng.uniform(0.01, 0.05, size=(P, N))
    betahat = rng.normal(size=(P, N)) * se * 1.5
    for p in range(P):
        betahat[p, missings[p]] = 0.0
        se[p, missings[p]] = 1.0
    var = np.geomspace(1e-7, 1e-2, M)
    covs = [v * (0.6 * np.eye(P) + 0.4 * np.ones((P, P))) for v in var]
    ann = np.zeros((N, A))
    cols = rng.integers(0, A - 1 if (empty_annot and A > 1) else A, size=N)
    ann[np.arange(N), cols] = 1
    return dict(P=P, N=N, M=M, perms=perms, missings=missings, blocks=blocks, se=se,
                betahat=betahat, covs=covs, ann=ann, t=ldthresh)


def _build(pr, which, **kw):
    common = dict(marginal_effects=pr['betahat'], std_errs=pr['se'], mixture_covs=pr['covs'],
                  annotations=pr['ann'], checkpoint=False, gwas_N=np.full(pr['P'], 5e4),
                  init_hg=np.full(pr['P'], 0.2), num_its=6, **kw)
    if which == 'oracle':
        from oracle.ldop import EigenBlock, BlockDiagonalLD
        from oracle.vi import MultiPopVIOracle
        ld = [BlockDiagonalLD([EigenBlock(X, pr['t']) for X in pr['blocks'][p]],
                              perm=pr['perms'][p], missing=pr['missings'][p])
              for p in range(pr['P'])]
        return MultiPopVIOracle(ld_mats=ld, **common)
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd.variational_inference import MultiPopVI
    ld = [BlockDiagonalMatrix([LowRankMatrix(X, pr['t']) for X in pr['blocks'][p]],
                              perm=pr['perms'][p], missing=pr['missings'][p])
          for p in range(pr['P'])]
    return MultiPopVI(ld_mats=ld, **common)


def _compare(pr, sweeps=4, **kw):
    ovi, vi = _build(pr, 'oracle', **kw), _build(pr, 'product', **kw)
    np.random.seed(7)
    op = ovi._initialize()
    np.random.seed(7)
    pp = vi._initialize()
    oe, pe = ovi.elbo(op), vi.elbo(pp)
    assert abs(oe - pe) <= 1e-8 * abs(oe) + 1e-9
    oL, pL, ored, pred = np.ones(5), np.ones(5), None, None
    for it in range(sweeps):
        op, oL, oe, ored = ovi._optimize_step(op, oL, oe, 2., ored)
        pp, pL, pe, pred = vi._optimize_step(pp, pL, pe, 2., pred)
        assert abs(oe - pe) <= 1e-8 * abs(oe) + 1e-9, (it, oe, pe)
        assert np.array_equal(oL, pL), (it, oL, pL)
    np.testing.assert_allclose(vi.real_posterior_mean(pp), ovi.real_posterior_mean(*op),
                               rtol=1e-6, atol=1e-11)
    np.testing.assert_allclose(pp[1], op[1], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(vi.error_scaling, ovi.error_scaling, rtol=1e-9)
    return vi


def test_minimal_sizes():
    rng = np.random.default_rng(1)
    _compare(_problem(rng, 1, [[1]], N=1, M=2))                       # one SNP, one 1x1 block
    _compare(_problem(rng, 1, [[1, 1, 2]], N=5, M=2))                 # 1-SNP blocks + a missing SNP
    _compare(_problem(rng, 2, [[3, 1], [2, 2]], N=4, M=3))


def test_cohort_without_any_ld():
    rng = np.random.default_rng(2)
    pr = _problem(rng, 2, [[40, 25], []], N=70, M=6)
    vi = _compare(pr)
    assert np.all(vi.ld_diags[1] == 0) and vi.ld_ranks[1] == 0


def test_annotation_nobody_has_and_learn_scaling():
    rng = np.random.default_rng(3)
    pr = _problem(rng, 1, [[60, 30]], N=100, M=8, A=3, empty_annot=True)
    vi = _compare(pr, scale_se=True, sweeps=6)
    assert vi.annotation_counts[2] == 0


def test_many_components_three_cohorts():
    rng = np.random.default_rng(4)
    _compare(_problem(rng, 2, [[50, 70], [120]], N=130, M=582), sweeps=3)   # CLI default M at P=2
    _compare(_problem(rng, 3, [[30, 40], [70], [20, 20, 30]], N=80, M=9), sweeps=3)


def test_five_to_eight_cohorts():
    """More than four cohorts: the reference takes any P through numpy's inv / slogdet
    (numerics.py:238-290, variational_inference.py:599-630); the per-SNP kernels are instantiated
    up to P = 8 (unrolled Cholesky).  P = 5 with different block partitions per cohort, annotations
    and --learn-scaling; P = 8 at the width limit; sweep by sweep against the oracle."""
    rng = np.random.default_rng(55)
    sizes5 = [[40, 30, 25], [60, 35], [95], [20, 20, 20, 20], [50, 45]]
    _compare(_problem(rng, 5, sizes5, N=110, M=7), sweeps=3)
    _compare(_problem(rng, 5, sizes5, N=110, M=6, A=2, ldthresh=0.8), sweeps=3, scale_se=True)
    sizes8 = [[30, 30], [60], [25, 35], [45], [20, 20, 20], [61], [33, 27], [50]]
    _compare(_problem(rng, 8, sizes8, N=70, M=5), sweeps=3)
    _compare(_problem(rng, 6, sizes8[:6], N=70, M=130), sweeps=2)      # many components, no stash


def test_big_blocks_and_thresholding():
    rng = np.random.default_rng(5)
    _compare(_problem(rng, 1, [[700, 130, 129]], N=1000, M=5), sweeps=3)
    _compare(_problem(rng, 2, [[300, 200], [257, 255]], N=520, M=5, ldthresh=0.5), sweeps=3,
             scaled=True)


def test_c5_shape_p4_m81():
    """The shape of BASELINE.json configs[4] -- 4 cohorts, M = 81 mixture components -- at a few
    thousand SNPs against the oracle: the Cholesky (P > 2) branch of every per-SNP kernel
    (reference numerics.py:238-290) with a large component count, multi-slab blocks, different
    partitions per cohort; plain and with --learn-scaling."""
    rng = np.random.default_rng(81)
    sizes = [[700, 300, 260], [520, 740], [129, 1131], [400, 400, 400, 60]]
    pr = _problem(rng, 4, sizes, N=1300, M=81)
    _compare(pr, sweeps=3)
    _compare(pr, sweeps=3, scale_se=True)
    pr = _problem(rng, 4, [[257, 300], [600], [128, 128, 300], [555]], N=640, M=81, A=2,
                  ldthresh=0.7)
    _compare(pr, sweeps=3, scaled=True)


def test_elbo_at_a_supplied_vi_delta():
    """MultiPopVI.elbo / real_posterior_* honour a vi_delta that is not the fixed point
    (vilma_eval_given_delta), P = 1, 2 and 4."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_driver_cpu import _supplied_delta_checks
    rng = np.random.default_rng(12)
    for P, sizes, N, M, A in ((1, [[130, 40]], 180, 7, 1),
                              (2, [[50, 70], [120]], 130, 6, 2),
                              (4, [[60], [30, 30], [129], [10, 200]], 230, 9, 1)):
        pr = _problem(rng, P, sizes, N=N, M=M, A=A)
        _supplied_delta_checks(_build(pr, 'product'), _build(pr, 'oracle'), 7)


@pytest.mark.parametrize('seed', range(10))
def test_random_configurations(seed):
    """Seeded random problems across the kernel template space (P, M, A, LD thresholds, block
    sizes on and around the 128-column slab and 256-column chunk boundaries, --scaled /
    --learn-scaling), three sweeps each against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    P = int(rng.integers(1, 5))
    M = int(rng.choice([2, 3, 7, 12, 33, 70]))
    A = int(rng.integers(1, 4))
    pool = [1, 2, 5, 31, 64, 127, 128, 129, 200, 255, 256, 257, 300, 513]
    sizes = [[int(v) for v in rng.choice(pool, size=int(rng.integers(1, 5)))] for _ in range(P)]
    N = max(sum(s) for s in sizes) + int(rng.integers(0, 9))
    pr = _problem(rng, P, sizes, N=N, M=M, A=A, ldthresh=float(rng.choice([1.0, 0.7])),
                  empty_annot=bool(rng.integers(0, 2)))
    _compare(pr, sweeps=3, scaled=bool(rng.integers(0, 2)), scale_se=bool(rng.integers(0, 2)))


@pytest.mark.parametrize('stored', [True, False])
@pytest.mark.parametrize('shape', ['p1', 'p2_a2', 'p3', 'p4_m81', 'p2_m130', 'p5', 'p2_scale_se',
                                   'p1_scale_se', 'p3_scale_se'])
def test_device_resident_sweeps_equal_host_decided_sweeps(shape, stored, monkeypatch):
    """The sweeps queued ahead (roles, L, step sizes on the device; decision kernel) against the
    same fit with every decision taken on the host side of the library (VILMA_LOOKAHEAD=0), on
    shapes the synthetic benchmark does not cover: one cohort, several annotations (the M-step
    of the decision kernel row by row), three and four cohorts (one candidate per trial: a
    rejected step makes the next queued trial run at twice the L), five cohorts (the general P x P
    Cholesky kernels, never a stash), many components (no stash: the trials store no vi_mu, the
    pass behind the decision forms the accepted candidate's responsibility sums and stores it).
    The *_scale_se shapes fit with --learn-scaling: the error-scaling update
    (_update_error_scaling, reference variational_inference.py:441-448, 472-486) and the
    re-evaluation behind it are decided and run on the device too.
    ELBO, L, error_scaling, convergence statistics after every sweep and the final state: bit for
    bit (to rounding where the trials store no vi_mu, see below); and the device handed no decision
    back to the host."""
    rng = np.random.default_rng({'p1': 1, 'p2_a2': 2, 'p3': 3, 'p4_m81': 4, 'p2_m130': 5,
                                 'p2_scale_se': 6, 'p1_scale_se': 7, 'p3_scale_se': 8, 'p5': 9}[shape])
    pr = {'p1': lambda: _problem(rng, 1, [[60, 45, 70, 30]], N=215, M=9),
          'p2_a2': lambda: _problem(rng, 2, [[80, 60, 40], [90, 50, 40]], N=190, M=8, A=3),
          'p3': lambda: _problem(rng, 3, [[50, 40], [45, 45], [30, 30, 30]], N=100, M=6, A=2),
          'p4_m81': lambda: _problem(rng, 4, [[70, 60], [130], [65, 65], [40, 40, 50]], N=140, M=81),
          'p2_m130': lambda: _problem(rng, 2, [[60, 50], [55, 55]], N=120, M=130),
          'p5': lambda: _problem(rng, 5, [[40, 30, 25], [60, 35], [95], [20, 20, 20, 20], [50, 45]],
                                 N=110, M=7),
          'p2_scale_se': lambda: _problem(rng, 2, [[80, 60, 40], [90, 50, 40]], N=190, M=8, A=2),
          'p1_scale_se': lambda: _problem(rng, 1, [[60, 45, 70, 30]], N=215, M=130),
          'p3_scale_se': lambda: _problem(rng, 3, [[50, 40], [45, 45], [30, 30, 30]], N=100, M=6)}[shape]()
    scale_se = shape.endswith('scale_se')
    # stored: trials of mixtures that fit the stash store their candidates (VILMA_STASH_LAZY=0: the form
    # of rounds 3 - 4) -- device-decided and
    # host-decided sweeps are then equal to the BIT; by default they run lazy trials that keep the stash
    # (late round 5) and are equal to rounding, every decision the same
    monkeypatch.setenv('VILMA_STASH_LAZY', '0' if stored else '1')

    def run(lookahead):
        monkeypatch.setenv('VILMA_LOOKAHEAD', '1' if lookahead else '0')
        vi = _build(pr, 'product', scale_se=scale_se)
        np.random.seed(3)
        vi._initialize()
        state, trace = None, []
        for k in range(16):
            state, stats = vi.sweep(state, lookahead=k < 15)
            trace.append((state['elbo'], tuple(state['L']), tuple(stats), tuple(vi.error_scaling)))
        params = vi._params()
        out = (trace, params[0].copy(), params[2].copy(), vi.n_trials, vi.n_stages_ahead,
               vi.n_stages_skipped)
        vi.engine.close()
        return out
    host = run(False)
    dev = run(True)
    if scale_se:
        assert any(t[3] != (1.0,) * pr['P'] for t in host[0])      # tau did get updated
    lazy = shape in ('p2_m130', 'p5', 'p1_scale_se') or not stored
    if lazy:
        # Mixtures beyond the stash: the queued sweeps' trials store no vi_mu and carry the beta loop's
        # state as mu_k = a mu_k^stored + Sig_k c (round 5), written out once when the loop ends; the
        # host-decided sweeps store every accepted candidate and blend the stored array again.  The
        # same fit up to the rounding of those intermediate arrays: every decision the same (L to
        # the bit, the same number of trials), values to 1e-12.
        for d, h in zip(dev[0], host[0]):
            assert d[1] == h[1]
            assert abs(d[0] - h[0]) <= 1e-12 * abs(h[0])
            assert d[2][0] == h[2][0]
            np.testing.assert_allclose(d[2], h[2], rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(d[3], h[3], rtol=1e-12)
        np.testing.assert_allclose(dev[1], host[1], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(dev[2], host[2], rtol=1e-10, atol=1e-300)
    elif scale_se:
        # with --learn-scaling the host-decided path forms the convergence statistics in a pass of
        # its own (vilma_mean_diff), the queued path inside the sweep's last evaluation: two
        # summation orders.  The count of moved means is exact either way; the rest to rounding.
        for d, h in zip(dev[0], host[0]):
            assert d[0] == h[0] and d[1] == h[1] and d[3] == h[3]
            assert d[2][0] == h[2][0]
            np.testing.assert_allclose(d[2], h[2], rtol=1e-12)
    else:
        assert dev[0] == host[0]
    if not lazy:
        assert np.array_equal(dev[1], host[1]) and np.array_equal(dev[2], host[2])
    assert dev[3] == host[3] and host[4] == 0
    # sweeps did run from the control block -- p2_m130 too: a mixture beyond the on-chip stash gets
    # its responsibility sums from a pass behind the decision, not from the host
    assert dev[4] >= 15 and dev[5] == 0


@pytest.mark.parametrize('two_step', ['1', '0'])
def test_learn_scaling_under_a_persistent_lazy_state(monkeypatch, two_step):
    """--learn-scaling with a mixture beyond the stash (M = 70, two cohorts): the state lives as
    (stored vi_mu, a, c) across sweeps, and a tau update changes Sig_k = (Prec_k + D / tau)^-1 under
    it -- the EVAL decision that takes the update has the state written out with the OLD tau (a pass
    queued beside the re-evaluation) and goes on from that array (decide.h).  30 sweeps in which tau
    moves again and again: the sweeps decided on the device against the host-decided fit (every
    decision equal, tau and ELBO to rounding) and, for the first sweeps, against the oracle."""
    rng = np.random.default_rng(77)
    pr = _problem(rng, 2, [[300, 129, 200], [257, 64, 513]], N=900, M=70, A=2)
    monkeypatch.setenv('VILMA_TWO_STEP', two_step)
    _compare(pr, sweeps=4, scale_se=True)

    def run(lookahead, persist):
        monkeypatch.setenv('VILMA_LOOKAHEAD', '1' if lookahead else '0')
        monkeypatch.setenv('VILMA_PIPE_PERSIST', '1' if persist else '0')
        vi = _build(pr, 'product', scale_se=True)
        np.random.seed(3)
        vi._initialize()
        state, trace = None, []
        for k in range(30):
            state, stats = vi.sweep(state, lookahead=k + 1 < 30)
            trace.append((state['elbo'], tuple(state['L']), tuple(vi.error_scaling)))
            if k == 20:
                vi.engine.drain()           # (a drain behind tau updates: the stored array has moved)
        out = (trace, vi._params()[0].copy(), vi.n_trials, vi.n_stages_ahead)
        vi.engine.close()
        return out
    host, stored, kept = run(False, False), run(True, False), run(True, True)
    moved = [k for k in range(1, 30) if host[0][k][2] != host[0][k - 1][2]]
    assert len(moved) >= 5, moved                       # tau did move, in many sweeps
    for other in (stored, kept):
        for (e_o, L_o, t_o), (e_h, L_h, t_h) in zip(other[0], host[0]):
            assert L_o == L_h
            assert abs(e_o - e_h) <= 1e-11 * abs(e_h)
            np.testing.assert_allclose(t_o, t_h, rtol=1e-11)
        np.testing.assert_allclose(other[1], host[1], rtol=1e-9, atol=1e-13)
        assert other[2] == host[2] and other[3] >= 20

