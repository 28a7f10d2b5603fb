"""GPU parity, fit level: the product's class API and CLI on the HIP engine against the
trajectories recorded from the reference and the reference's own golden files.

Bars (north star): SNP->block assignment bit exact; ELBO trajectory and posterior means within
1e-5 relative -- the assertions here are far tighter (1e-9 / 1e-7) because everything is fp64."""
import os
import sys

import numpy as np
import pytest

from helpers import golden, product_vi_from_traj, check_trajectory, TRAJ_NAMES, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', TRAJ_NAMES)
def test_fit_trajectory(name):
    g = golden('traj_%s.npz' % name)
    vi, ld = product_vi_from_traj(g)
    from vilma_amd.engine import HipEngine
    assert isinstance(vi.engine, HipEngine)
    check_trajectory(vi, g)


@pytest.mark.parametrize('lookahead', [False, True])
def test_mid_size_trajectory_with_inner_loops(lookahead):
    """5 000 LD SNPs x 2 cohorts, 16 sweeps recorded from the reference, among them sweeps with
    several beta updates, rejected first steps and a trial whose two candidates are BOTH rejected.
    lookahead=True: through the device-resident path (sweeps queued ahead and decided on the
    device); lookahead=False: every decision taken by the library's host code.  Same bars either
    way: L to the bit, ELBO 1e-9, posterior means 1e-7."""
    from helpers import check_compact_trajectory
    g = golden('traj_p2_mid.npz')
    assert int(g['objs_per_sweep'].max()) >= 9 and g['L'][0, 0] >= 4.0
    vi, _ = product_vi_from_traj(g)
    counts = check_compact_trajectory(vi, g, lookahead=lookahead)
    assert max(t for _, t in counts) >= 3          # a sweep with three or more beta trials
    if lookahead:
        # every sweep but the last one's tail ran from the control block: the inner beta loops and
        # the both-candidates-rejected trial were decided on the device, nothing went to the host
        assert vi.n_stages_ahead >= 15 and vi.n_stages_skipped == 0


@pytest.mark.parametrize('sums', ['stash', 'pass', 'lazy', 'lazy-one-step', 'lazy-stash'])
def test_inner_loops_on_the_device_equal_host_decided_bit_for_bit(monkeypatch, sums):
    """The mid-size problem's 16 sweeps (inner beta loops of up to eight updates, rejected steps, a
    trial with both candidates rejected) decided on the device against the same fit with every
    decision taken by the library's host code: ELBO, L, running change, convergence statistics per
    sweep and the final vi_mu bit for bit.  Three ways the queued path gets the accepted
    candidate's responsibility sums: from the trial pass's on-chip stash (what M = 20 takes by
    itself); from a pass over the accepted candidate behind the decision (what mixtures beyond the
    stash take: forced here); and that pass behind LAZY trials, which store no vi_mu at all -- the
    state of the beta loop is carried as two numbers per SNP and written out when the loop ends (also
    with one candidate per trial, VILMA_TWO_STEP=0: what more than two cohorts take; these two
    variants are compared to rounding, see below)."""
    g = golden('traj_p2_mid.npz')
    # (the stash is switched off for the host-decided run too: the two ways of summing the same
    # responsibilities differ in the last bit, and that is not what is compared here)
    # ('lazy-stash', late round 5: lazy trials that keep the on-chip stash -- what this mixture takes by
    # default without --learn-scaling)
    monkeypatch.setenv('VILMA_TILE_SUMS', '1' if sums in ('stash', 'lazy-stash') else '0')
    monkeypatch.setenv('VILMA_PIPE_LAZY', '1' if sums.startswith('lazy') else '0')
    if sums == 'lazy-one-step':
        monkeypatch.setenv('VILMA_TWO_STEP', '0')

    def run(lookahead):
        monkeypatch.setenv('VILMA_LOOKAHEAD', '1' if lookahead else '0')
        vi, _ = product_vi_from_traj(g)
        np.random.seed(int(g['seed']))
        vi._initialize()
        state, trace = None, []
        for k in range(len(g['elbo'])):
            state, stats = vi.sweep(state, lookahead=k + 1 < len(g['elbo']))
            trace.append((state['elbo'], tuple(state['L']), state['running'], tuple(stats)))
        out = (trace, vi._params()[0].copy(), vi.n_trials, vi.n_evaluations, vi.n_stages_ahead,
               vi.n_stages_skipped)
        vi.engine.close()
        return out
    host, dev = run(False), run(True)
    if sums.startswith('lazy'):
        # Round 5: between two stored states lazy trials carry the beta loop's state as (a, c) --
        # mu_k = a mu_k^stored + Sig_k c -- and write it out once, when the loop ends; the host's line
        # search stores every accepted candidate and blends the stored (rounded) array again.  The
        # same numbers up to the rounding of the intermediate vi_mu arrays: every decision the same
        # (L to the bit, same trials, same points), values to 1e-12.
        for (e_d, L_d, r_d, st_d), (e_h, L_h, r_h, st_h) in zip(dev[0], host[0]):
            assert L_d == L_h
            assert abs(e_d - e_h) <= 1e-12 * abs(e_h) and abs(r_d - r_h) <= 1e-9 * abs(r_h)
            np.testing.assert_allclose(st_d, st_h, rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(dev[1], host[1], rtol=1e-10, atol=1e-14)
    else:
        assert dev[0] == host[0]
        assert np.array_equal(dev[1], host[1])
    assert dev[2] == host[2] and dev[3] == host[3]
    assert host[4] == 0 and dev[4] >= 15 and dev[5] == 0


@pytest.mark.parametrize('two_step', ['1', '0'])
def test_persistent_lazy_state_against_a_stored_one_and_through_drains(monkeypatch, two_step):
    """Mixtures beyond the stash, no --learn-scaling: the state lives as (stored vi_mu, a, c) from
    sweep to sweep -- no sweep writes vi_mu, the sums pass only sums, the evaluation behind the
    M-step derives its state as the trials do (SweepCtl::mu_base).  Against the same fit with the
    state written out at the end of every sweep (VILMA_PIPE_PERSIST=0) and against the host-decided
    fit: every decision the same (L to the bit), values to rounding.  And the state comes back as an
    array whenever somebody asks: a drain between two queued sweeps (the reported state's c must
    have survived the trial queued beyond it), a parameter read in mid-fit, the end of the run."""
    g = golden('traj_p2_mid.npz')
    monkeypatch.setenv('VILMA_TILE_SUMS', '0')
    monkeypatch.setenv('VILMA_PIPE_LAZY', '1')
    monkeypatch.setenv('VILMA_TWO_STEP', two_step)
    n = len(g['elbo'])

    def run(lookahead, persist, poke=False, n=n, pure=True):
        monkeypatch.setenv('VILMA_LOOKAHEAD', '1' if lookahead else '0')
        monkeypatch.setenv('VILMA_PIPE_PERSIST', '1' if persist else '0')
        # (the state _initialize builds is mu_k = Sig_k c: by default the persistent form starts with
        # a = 0 and no pass reads a vi_mu array; VILMA_PURE_START=0 starts from the stored array)
        monkeypatch.setenv('VILMA_PURE_START', '1' if pure else '0')
        vi, _ = product_vi_from_traj(g)
        np.random.seed(int(g['seed']))
        vi._initialize()
        state, trace, seen = None, [], []
        for k in range(n):
            state, stats = vi.sweep(state, lookahead=k + 1 < n)
            trace.append((state['elbo'], tuple(state['L']), state['running']))
            if poke and k == 4:
                vi.engine.drain()                       # the promise of another sweep broken
            if poke and k in (8, 9):
                seen.append((k, vi._params()[0].copy()))    # a state read in mid-fit (twice in a row)
        form = vi.engine.state_form()
        out = (trace, vi._params()[0].copy(), vi.n_trials, vi.n_stages_ahead, seen, form)
        vi.engine.close()
        return out
    host = run(False, False)
    stored = run(True, False)
    kept = run(True, True)
    poked = run(True, True, poke=True)
    based = run(True, True, pure=False)
    # the form the queued sweeps held the state in: base-free by default (through the drain and the
    # state reads too: a write-out re-establishes it), (stored vi_mu, a, c) without the pure start,
    # a stored array when every sweep writes it out
    assert kept[5] == 2 and poked[5] == 2 and based[5] == 1 and stored[5] == 0 and host[5] == 0
    for other in (stored, kept, poked, based):
        for (e_o, L_o, r_o), (e_h, L_h, r_h) in zip(other[0], host[0]):
            assert L_o == L_h
            assert abs(e_o - e_h) <= 1e-12 * abs(e_h) and abs(r_o - r_h) <= 1e-9 * abs(r_h)
        np.testing.assert_allclose(other[1], host[1], rtol=1e-10, atol=1e-14)
        assert other[2] == host[2]
    assert kept[3] >= 15 and poked[3] >= 10
    # the arrays handed out in mid-fit are states of the trajectory: the next sweeps go on from them
    # (the uninterrupted runs above end where the interrupted one does), and they differ from sweep
    # to sweep
    (k0, m0), (k1, m1) = poked[4]
    assert np.all(np.isfinite(m0)) and np.all(np.isfinite(m1)) and not np.array_equal(m0, m1)
    # ... and the array read after sweep 8 is what a host-decided fit of nine sweeps ends with
    np.testing.assert_allclose(m0, run(False, False, n=9)[1], rtol=1e-10, atol=1e-14)


def test_changing_the_stream_between_queued_sweeps(monkeypatch):
    """A fit that moves to another HIP stream while sweeps are queued ahead on the old one: the
    library finishes what it queued there, puts the reported state back and carries on on the new
    stream -- the same trajectory as a fit that never moved (to rounding: the moments of the
    restored state are re-derived)."""
    import torch
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS
    from vilma_amd.engine import HipEngine
    from vilma_amd.sharding import Comm
    from vilma_amd.variational_inference import SweepDriver
    device = torch.device('cuda', 0)

    def run(switch):
        sh = SyntheticShard(seed=5, **WORKLOADS['tiny']).build(device)
        sh.finish_init(sh.inv_se2_local)
        eng = HipEngine(sh.P, sh.N, sh.M, 1)
        eng.set_snp_data(sh.adj, sh.se, sh.sld, sh.scalings, sh.annot)
        eng.set_mixture(np.linalg.inv(sh.covs), np.linalg.slogdet(sh.covs)[1])
        for p in range(sh.P):
            eng.load_ld(p, sh.ld_blocks_torch(p, device), sh.perm, sh.n_ld, specs=sh.block_specs())
        drv = SweepDriver()
        drv._setup_driver(eng, Comm(), sh.P, sh.M, 1, sh.chi_local, sh.rank_local, [sh.N_global],
                          np.linalg.slogdet(sh.covs)[1], scale_se=False, num_its=100)
        drv.initialize_from(sh.fake_mu)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        state, trace = None, []
        for k in range(12):
            which = streams[(k // 4) % 2] if switch else streams[0]
            with torch.cuda.stream(which):
                eng.refresh_stream()
                state, stats = drv.sweep(state, lookahead=k < 11)
            trace.append((state['elbo'], tuple(state['L'])))
        torch.cuda.synchronize()
        mu = eng.get_mu()
        eng.close()
        return trace, mu
    one, mu_one = run(False)
    two, mu_two = run(True)
    for a, b in zip(two, one):
        assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and a[1] == b[1]
    np.testing.assert_allclose(mu_two, mu_one, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize('form', ['dense', 'eig'])
@pytest.mark.parametrize('name', ['p2_lowrank', 'p4_general', 'p2_bigblock_lr'])
def test_fit_trajectory_forms(name, form):
    g = golden('traj_%s.npz' % name)
    vi, _ = product_vi_from_traj(g, form=form)
    check_trajectory(vi, g)


@pytest.mark.parametrize('name', ['p1_dense', 'p2_scale_se'])
def test_optimize_converges_like_reference(name):
    g = golden('traj_%s.npz' % name)
    cap = {'p1_dense': 40, 'p2_scale_se': 30}[name]
    vi, _ = product_vi_from_traj(g, num_its=cap)
    np.random.seed(int(g['seed']))
    params = vi.optimize()
    assert vi.num_its_run == int(g['opt_num_its'])
    np.testing.assert_allclose(vi.real_posterior_mean(params), g['opt_post_mean'], rtol=1e-6,
                               atol=1e-12)
    np.testing.assert_allclose(vi.error_scaling, g['opt_error_scaling'], rtol=1e-8)


def test_cli_golden_runs(tmp_path):
    """reference tests/test.py:2161-2197 and example/example.sh + checkpoint_example.sh."""
    import test_cli_cpu as cli
    cli.fit_test_cli(tmp_path, None)
    cli.fit_test_cli(tmp_path, None, manifest='ld_manifest_svd.tsv')
    cli.fit_example(tmp_path, None)


def test_block_diagonal_dot_on_gpu():
    """BlockDiagonalMatrix.dot (the product class) == the reference operator on the loader
    fixtures, incl. flips (R[0,2] = -1) and zero rows at missing (tests/test.py:594-706)."""
    from vilma_amd import load
    K = golden('loader_kat.npz')
    ref = os.path.join(GOLDEN, 'refdata')
    cases = {'plain': ('ld_manifest.tsv', 'good_variants.tsv', [], 1.0),
             'thresh': ('ld_manifest.tsv', 'good_variants.tsv', [], 0.8),
             'deny': ('ld_manifest.tsv', 'good_variants.tsv', [3, 4, 5], 1.0),
             'svd_deny': ('ld_manifest_svd.tsv', 'good_variants.tsv', [3, 4, 5], 0.8),
             'plusmissing': ('ld_manifest.tsv', 'good_variants_plus_missing.tsv', [], 1.0)}
    for tag, (manifest, varfile, deny, t) in cases.items():
        variants = load.load_variant_list(os.path.join(ref, varfile))
        bd, missing = load.load_ld_from_schema(os.path.join(ref, manifest), variants, deny, t)
        v = np.linspace(-1, 1, bd.shape[0])
        np.testing.assert_allclose(bd.dot(v), K[tag + '_dot'], atol=1e-12, err_msg=tag)
        for i in missing:
            e = np.zeros(bd.shape[0]); e[i] = 1
            assert np.all(bd.dot(e) == 0)
    variants = load.load_variant_list(os.path.join(ref, 'good_variants.tsv'))
    bd, _ = load.load_ld_from_schema(os.path.join(ref, 'ld_manifest.tsv'), variants, [], 1.)
    want = np.eye(13); want[0, 2] = want[2, 0] = -1; want[5, 5] = want[12, 12] = 0
    v = np.random.default_rng(0).random(13)
    np.testing.assert_allclose(bd.dot(v), want.dot(v), atol=1e-12)


def _rank_main(rank, world, port, name, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                            world_size=world)
    try:
        g = golden('traj_%s.npz' % name)
        vi, _ = product_vi_from_traj(g)          # HIP engine on cuda:0, sums reduced over gloo
        assert vi.comm.world == world
        check_trajectory(vi, g)
        q.put((rank, 'ok', len(vi._snps)))
    except BaseException:                          # noqa: BLE001
        import traceback
        q.put((rank, 'fail', traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu():
    """Shard-count invariance with the real engine: 2 processes (both on cuda:0), LD blocks
    split between them, reproduce the reference trajectory of the unsharded problem."""
    import torch.multiprocessing as mp
    name = 'p2_scale_se'
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in results:
        assert status == 'ok', 'rank %d failed:\n%s' % (rank, info)
    assert sum(info for _, _, info in results) == int(golden('traj_%s.npz' % name)['N'])


def test_synthetic_closed_form_constants_match_class_api():
    """bench.py enters below the class API with closed-form load-time constants (AR(1) LD is
    full rank); on a small instance they equal what the class API derives through eigh."""
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS, ar1_numpy
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd.variational_inference import MultiPopVI
    cfg = dict(WORKLOADS['tiny'])
    sh = SyntheticShard(seed=3, **cfg).build(None)
    sh.finish_init(sh.inv_se2_local)
    ld = [BlockDiagonalMatrix([LowRankMatrix(ar1_numpy(b.n, b.rho[p]), 1.0) for b in sh.blocks],
                              perm=sh.perm, missing=sh.missing) for p in range(sh.P)]
    vi = MultiPopVI(marginal_effects=sh.betahat, std_errs=sh.se, ld_mats=ld,
                    mixture_covs=list(sh.covs), annotations=np.ones((sh.N, 1)), checkpoint=False,
                    gwas_N=sh.gwas_N, init_hg=sh.init_hg, num_its=5)
    np.testing.assert_allclose(vi.adj_marginal_effects, sh.adj, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(vi.chi_stat, sh.chi_local, rtol=1e-8)
    np.testing.assert_allclose(vi.ld_ranks, sh.rank_local)
    np.testing.assert_allclose(vi.inverse_betas, sh.inverse_betas, rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(vi.ld_diags, sh.ld_diags, atol=1e-12)


def test_synthetic_eigen_form_constants_match_class_api():
    """The C4-style workload gives LD directly as (U, s); its closed-form constants equal what the
    class API derives from LowRankMatrix(u, s, v, D=0), and both LD storage forms fit alike."""
    import torch
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd.variational_inference import MultiPopVI
    dev = torch.device('cuda', 0)
    sh = SyntheticShard(seed=4, **WORKLOADS['tiny4']).build(dev)
    sh.finish_init(sh.inv_se2_local)
    ld = []
    for p in range(sh.P):
        mats = []
        for U, sv in sh._eig[p]:
            u = U.cpu().numpy()
            mats.append(LowRankMatrix(u=u, s=sv.cpu().numpy(), v=u.T.copy(), D=np.zeros(u.shape[0])))
        ld.append(BlockDiagonalMatrix(mats, perm=sh.perm, missing=sh.missing))
    elbos = {}
    for form in ('eig', 'dense'):
        vi = MultiPopVI(marginal_effects=sh.betahat, std_errs=sh.se, ld_mats=ld,
                        mixture_covs=list(sh.covs), annotations=np.ones((sh.N, 1)),
                        checkpoint=False, gwas_N=sh.gwas_N, init_hg=sh.init_hg, num_its=3, form=form)
        np.testing.assert_allclose(vi.adj_marginal_effects, sh.adj, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(vi.chi_stat, sh.chi_local, rtol=1e-8)
        np.testing.assert_allclose(vi.ld_ranks, sh.rank_local)
        np.testing.assert_allclose(vi.ld_diags, sh.ld_diags, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(vi.inverse_betas, sh.inverse_betas, rtol=1e-6, atol=1e-10)
        np.random.seed(1)
        vi.optimize()
        elbos[form] = vi._objective
    assert abs(elbos['eig'] - elbos['dense']) < 1e-9 * abs(elbos['eig'])


def test_factor_model_workload_is_what_the_loader_makes_of_the_same_matrices():
    """C4f follows SURVEY 8d's C4 recipe: R = D^-1/2 (F F^T/m + 0.05 I) D^-1/2 cut at --ldthresh
    0.8.  Rebuild every block's R on the host from the same random stream, hand it to the
    product's own loader type (LowRankMatrix(X, t=0.8): host LAPACK + the reference's selection
    rule) and compare ranks, operators and the class API's load-time constants."""
    import torch
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS, FACTOR_LD_THRESH
    from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
    from vilma_amd.variational_inference import MultiPopVI
    dev = torch.device('cuda', 0)
    sh = SyntheticShard(seed=4, **WORKLOADS['tiny4f']).build(dev)
    sh.finish_init(sh.inv_se2_local)
    ld = []
    for p in range(sh.P):
        mats = []
        for i, blk in enumerate(sh.blocks):
            m = -(-blk.n // 4)
            rng = np.random.default_rng([sh.seed, 5000 + sh.b0 + i, p])
            F = rng.normal(size=(blk.n, m))
            R = F @ F.T / m + 0.05 * np.eye(blk.n)
            d = 1.0 / np.sqrt(np.diag(R))
            R = d[:, None] * R * d[None, :]
            mat = LowRankMatrix(X=R, t=FACTOR_LD_THRESH)
            U, sv = sh._eig[p][i]
            assert mat.s.shape[0] == U.shape[1] == sh.ranks_by_cohort[p][i] >= m == sh.ranks[i]
            got = ((U * sv) @ U.T).cpu().numpy()
            np.testing.assert_allclose(got, (mat.u * mat.s) @ mat.v, rtol=0, atol=1e-11)
            mats.append(mat)
        ld.append(BlockDiagonalMatrix(mats, perm=sh.perm, missing=sh.missing))
    elbos = {}
    for form in ('eig', 'dense'):
        vi = MultiPopVI(marginal_effects=sh.betahat, std_errs=sh.se, ld_mats=ld,
                        mixture_covs=list(sh.covs), annotations=np.ones((sh.N, 1)),
                        checkpoint=False, gwas_N=sh.gwas_N, init_hg=sh.init_hg, num_its=3, form=form)
        np.testing.assert_allclose(vi.adj_marginal_effects, sh.adj, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(vi.chi_stat, sh.chi_local, rtol=1e-8)
        np.testing.assert_allclose(vi.ld_ranks, sh.rank_local)
        np.testing.assert_allclose(vi.ld_diags, sh.ld_diags, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(vi.inverse_betas, sh.inverse_betas, rtol=1e-6, atol=1e-10)
        np.random.seed(1)
        vi.optimize()
        elbos[form] = vi._objective
    assert abs(elbos['eig'] - elbos['dense']) < 1e-9 * abs(elbos['eig'])


def test_cli_under_torchrun_two_ranks(tmp_path):
    """`torchrun --nproc-per-node 2 -m vilma_amd fit ...` (rehearsed over gloo with both ranks on
    this GPU): rank 0 writes the same outputs as the single-process run."""
    import subprocess
    ref = os.path.join(GOLDEN, 'refdata')
    common = ['fit', '--ld-schema', os.path.join(ref, 'ld_manifest.tsv'),
              '--sumstats', os.path.join(ref, 'good_sumstats_beta.tsv'),
              '-K', '20', '--ldthresh', '0.8', '--init-hg', '0.2', '--samplesizes', '10e3',
              '--names', 'c', '--learn-scaling', '--num-its', '12',
              '--extract', os.path.join(ref, 'good_variants.tsv')]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    one = str(tmp_path / 'one')
    subprocess.check_call([sys.executable, '-m', 'vilma_amd'] + common + ['--output', one],
                          env=env, cwd=root)
    two = str(tmp_path / 'two')
    env2 = dict(env, VILMA_DIST_BACKEND='gloo', VILMA_SAME_DEVICE='1')
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                           '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                           '--master-port', str(29700 + os.getpid() % 200), '-m', 'vilma_amd']
                          + common + ['--output', two], env=env2, cwd=root)
    a, b = np.load(one + '.npz'), np.load(two + '.npz')
    for key in a.files:
        np.testing.assert_allclose(b[key], a[key], rtol=1e-9, atol=1e-13, err_msg=key)
    ta = open(one + '.estimates.tsv').read().splitlines()
    tb = open(two + '.estimates.tsv').read().splitlines()
    assert ta[0] == tb[0] and len(ta) == len(tb)


def test_side_stream_does_not_change_a_bit(tmp_path):
    """The side stream only changes WHEN the responsibility sums and convergence statistics run,
    not what they compute: the fit with VILMA_OVERLAP=0 is bit-identical."""
    import subprocess
    ex = os.path.join(GOLDEN, 'example')
    common = ['fit', '--sumstats', os.path.join(ex, 'example_data', 'example_gwas_sumstats.txt'),
              '--ld-schema', os.path.join(ex, 'ld_mat', 'example_schema.schema'),
              '--seed', '42', '-K', '81', '--init-hg', '0.2', '--samplesizes', '300e3',
              '--names', 'ukbb', '--learn-scaling', '--num-its', '25',
              '--extract', os.path.join(ex, 'keep_variants.txt')]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for flag in ('1', '0'):
        outs[flag] = str(tmp_path / ('overlap' + flag))
        subprocess.check_call([sys.executable, '-m', 'vilma_amd'] + common + ['--output', outs[flag]],
                              env=dict(os.environ, PYTHONPATH=root, VILMA_OVERLAP=flag), cwd=root)
    a, b = np.load(outs['1'] + '.npz'), np.load(outs['0'] + '.npz')
    for key in a.files:
        assert np.array_equal(a[key], b[key]), key
    assert open(outs['1'] + '.estimates.tsv').read() == open(outs['0'] + '.estimates.tsv').read()


def test_rccl_stream_ordering_with_one_rank_group():
    """A one-rank nccl (= RCCL) process group with forced collectives: every decision of the fit
    goes kernels -> RCCL all-reduce on a slice of the result vector -> pinned fetch, exactly the
    multi-GPU sequence, and must reproduce the reference trajectory."""
    import torch
    import torch.distributed as dist
    from vilma_amd.sharding import Comm
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ['MASTER_PORT'] = str(29800 + os.getpid() % 100)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        comm = Comm(force=True)
        assert comm.active and comm.backend == 'nccl'
        for name in ('p2_scale_se', 'p2_lowrank'):
            g = golden('traj_%s.npz' % name)
            vi, _ = product_vi_from_traj(g, comm=comm)
            check_trajectory(vi, g)
            np.random.seed(int(g['seed']))
            vi2, _ = product_vi_from_traj(g, comm=comm, num_its=12)
            vi2.optimize()                       # the pipelined loop under RCCL ordering
            vi3, _ = product_vi_from_traj(g, num_its=12)
            np.random.seed(int(g['seed']))
            vi3.optimize()
            assert vi2.num_its_run == vi3.num_its_run
            assert abs(vi2._objective - vi3._objective) <= 1e-12 * abs(vi3._objective)
    finally:
        dist.destroy_process_group()


def _sweep_trace(monkeypatch, two_step, lookahead, n_sweeps=35):
    """ELBO / L after every sweep of the synthetic 'tiny' problem (2 cohorts, 6.3 k SNPs, blocks
    up to ~1 000 SNPs) driven like bench.py drives it."""
    import torch
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS
    from vilma_amd.engine import HipEngine
    from vilma_amd.sharding import Comm
    from vilma_amd.variational_inference import SweepDriver
    monkeypatch.setenv('VILMA_TWO_STEP', '1' if two_step else '0')
    monkeypatch.setenv('VILMA_LOOKAHEAD', '1' if lookahead else '0')
    device = torch.device('cuda', 0)
    sh = SyntheticShard(seed=3, **WORKLOADS['tiny']).build(device)
    sh.finish_init(sh.inv_se2_local)
    eng = HipEngine(sh.P, sh.N, sh.M, 1)
    eng.set_snp_data(sh.adj, sh.se, sh.sld, sh.scalings, sh.annot)
    eng.set_mixture(np.linalg.inv(sh.covs), np.linalg.slogdet(sh.covs)[1])
    for p in range(sh.P):
        eng.load_ld(p, sh.ld_blocks_torch(p, device), sh.perm, sh.n_ld, specs=sh.block_specs())
    drv = SweepDriver()
    drv._setup_driver(eng, Comm(), sh.P, sh.M, 1, sh.chi_local, sh.rank_local, [sh.N_global],
                      np.linalg.slogdet(sh.covs)[1], scale_se=False, num_its=100)
    drv.initialize_from(sh.fake_mu)
    state, trace = None, []
    for k in range(n_sweeps):
        state, stats = drv.sweep(state, lookahead=k + 1 < n_sweeps)
        trace.append((state['elbo'], tuple(state['L']), tuple(stats)))
    mu = eng.get_mu()
    counts = (drv.n_evaluations, drv.n_trials, drv.n_products, drv.n_stages_ahead)
    eng.close()
    return trace, mu, counts


def test_reading_the_state_between_sweeps_queued_ahead(monkeypatch):
    """A caller that promises another sweep (lookahead) and then reads the state instead gets the
    state of the sweep that was REPORTED: the library waits for the sweep it queued ahead and puts
    the reported state back (the vi_mu is still in its buffer, the moments are re-derived), and the
    fit goes on from there like one that never looked (same L trajectory; the re-derived moments
    may differ from the originals in the last bit, which shows in the ELBO and the convergence
    statistics at 1e-15)."""
    import torch
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS
    from vilma_amd.engine import HipEngine
    from vilma_amd.sharding import Comm
    from vilma_amd.variational_inference import SweepDriver
    device = torch.device('cuda', 0)

    def run(peek_at):
        sh = SyntheticShard(seed=3, **WORKLOADS['tiny']).build(device)
        sh.finish_init(sh.inv_se2_local)
        eng = HipEngine(sh.P, sh.N, sh.M, 1)
        eng.set_snp_data(sh.adj, sh.se, sh.sld, sh.scalings, sh.annot)
        eng.set_mixture(np.linalg.inv(sh.covs), np.linalg.slogdet(sh.covs)[1])
        for p in range(sh.P):
            eng.load_ld(p, sh.ld_blocks_torch(p, device), sh.perm, sh.n_ld, specs=sh.block_specs())
        drv = SweepDriver()
        drv._setup_driver(eng, Comm(), sh.P, sh.M, 1, sh.chi_local, sh.rank_local, [sh.N_global],
                          np.linalg.slogdet(sh.covs)[1], scale_se=False, num_its=100)
        drv.initialize_from(sh.fake_mu)
        state, trace, peeks = None, [], {}
        for k in range(14):
            state, stats = drv.sweep(state, lookahead=True)
            trace.append((state['elbo'], tuple(state['L']), tuple(stats)))
            if k in peek_at:
                params = drv._params()
                peeks[k] = (params[0].copy(), params[1].copy(), eng.get_moments()[0].copy(),
                            drv._objective)
        mu = eng.get_mu()
        eng.close()
        return trace, peeks, mu
    plain, _, mu_plain = run(())
    peeked, peeks, mu_peeked = run((3, 4, 9))
    for a, b in zip(peeked, plain):
        assert abs(a[0] - b[0]) <= 1e-13 * abs(b[0]) and a[1] == b[1]
        np.testing.assert_allclose(a[2], b[2], rtol=1e-8)     # (the max RELATIVE change divides by ~0)
    np.testing.assert_allclose(mu_peeked, mu_plain, rtol=1e-9, atol=1e-300)
    # what was read at sweep k is the state after sweep k: the same as a run that stops there
    for k, (vi_mu, vi_delta, mean, obj) in peeks.items():
        assert obj == plain[k][0] or abs(obj - plain[k][0]) < 1e-9 * abs(obj)
        assert np.all(np.isfinite(vi_mu)) and np.all(np.isfinite(vi_delta)) and np.all(np.isfinite(mean))
    stop, peeks4, _ = run((4,))
    np.testing.assert_allclose(peeks4[4][0], peeks[4][0], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize('stored', [True, False])
def test_two_step_and_lookahead_change_no_bit(monkeypatch, stored):
    """Two line-search steps per device pass and the stage queued ahead of the device's decision
    are pure scheduling: ELBO, L, convergence statistics after every sweep and the final vi_mu are
    BIT-identical to the one-step, host-decides-everything schedule -- as long as the trials store
    their candidates (VILMA_STASH_LAZY=0; what fits with --learn-scaling do by default).  By default
    the sweeps queued ahead run lazy trials (the state carried as (stored vi_mu, a, c): no vi_mu
    array is stored and blended again) and are equal to rounding, every decision the same."""
    monkeypatch.setenv('VILMA_STASH_LAZY', '0' if stored else '1')
    base, mu0, c0 = _sweep_trace(monkeypatch, two_step=False, lookahead=False)
    assert c0[2] == c0[0] and c0[3] == 0                 # one LD pass per point, nothing queued ahead
    for two_step, lookahead in ((True, False), (False, True), (True, True)):
        trace, mu, c = _sweep_trace(monkeypatch, two_step, lookahead)
        if stored or not lookahead:
            assert trace == base, (two_step, lookahead)
            assert np.array_equal(mu, mu0)
        else:
            for (e, L, st), (e0, L0, st0) in zip(trace, base):
                assert L == L0 and st[0] == st0[0]
                assert abs(e - e0) <= 1e-12 * abs(e0)
                np.testing.assert_allclose(st, st0, rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(mu, mu0, rtol=1e-10, atol=1e-14)
        assert c[:2] == c0[:2]                           # same points looked at, same trials
        if two_step:
            assert c[2] < c0[2]                          # ... in fewer passes over the LD store
        if lookahead:
            assert c[3] > 0


@pytest.mark.parametrize('name', ['p1_dense', 'p2_scale_se', 'p1_scaled', 'p4_general'])
def test_output_arrays_formed_on_the_device(name):
    """The arrays `vilma fit` writes besides vi_mu (reference vi_options.py:263-265): vi_sigma
    [M,P,P,N] from vilma_get_vi_sigma against the host's closed forms (one and two cohorts: the
    same IEEE operations, equal to the bit; more: Cholesky against numpy's inverse, 1e-12) at an
    error scaling that has moved; vi_delta [N,M] (transposed on the device) is compared with the
    reference's fixed point of the same vi_mu in test_fit_trajectory (here: its layout and normalisation)."""
    g = golden('traj_%s.npz' % name)
    vi, _ = product_vi_from_traj(g)
    np.random.seed(int(g['seed']))
    params = vi._initialize()
    L, elbo, red = np.ones(5), vi.elbo(params), None
    for it in range(3):
        params, L, elbo, red = vi._optimize_step(params, L, elbo, 2., red)
    vi_delta = params[1]                       # (downloaded at the fit's own error scaling)
    if bool(g['scale_se']):
        vi.error_scaling = vi.error_scaling * np.linspace(1.1, 0.9, vi.num_pops)    # off 1 either way
    dev, host = vi.vi_sigma, vi._vi_sigma_host()
    assert dev.shape == host.shape == (vi.num_mix, vi.num_pops, vi.num_pops, vi.num_loci)
    if vi.num_pops <= 2:
        assert np.array_equal(dev, host)
    else:
        np.testing.assert_allclose(dev, host, rtol=1e-12, atol=1e-300)
    assert vi_delta.shape == (vi.num_loci, vi.num_mix) and vi_delta.flags.c_contiguous
    np.testing.assert_allclose(vi_delta.sum(axis=1), 1.0, rtol=1e-12)


@pytest.mark.parametrize('kill', [1, 3])
def test_a_lazy_beta_loop_handed_back_to_the_host_mid_way(monkeypatch, kill):
    """Lazy trials carry the beta loop's state as (stored vi_mu, a, c) and nothing of size [M][P][N]
    is written until the loop ends.  If the device hands a sweep back to the host in the middle of
    such a loop (VILMA_DEBUG_KILL_DEFERRED=k: the k-th trial decision that finds the state in that
    form refuses to decide; in a real fit: a line search beyond L_MAX), the host writes the state
    out and its own line search -- which works on stored vi_mu -- finishes the sweep: the fit equals
    the host-decided one (every decision, values to rounding)."""
    g = golden('traj_p2_mid.npz')
    monkeypatch.setenv('VILMA_TILE_SUMS', '0')
    monkeypatch.setenv('VILMA_PIPE_LAZY', '1')

    def run(lookahead, kill_at):
        monkeypatch.setenv('VILMA_LOOKAHEAD', '1' if lookahead else '0')
        monkeypatch.setenv('VILMA_DEBUG_KILL_DEFERRED', str(kill_at))
        vi, _ = product_vi_from_traj(g)
        np.random.seed(int(g['seed']))
        vi._initialize()
        state, trace = None, []
        for k in range(len(g['elbo'])):
            state, stats = vi.sweep(state, lookahead=k + 1 < len(g['elbo']))
            trace.append((state['elbo'], tuple(state['L']), state['running']))
        out = (trace, vi._params()[0].copy(), vi.n_trials, vi.n_stages_skipped)
        vi.engine.close()
        return out
    host, dev = run(False, 0), run(True, kill)
    assert dev[3] >= 1                                  # a sweep did come back to the host
    for (e_d, L_d, r_d), (e_h, L_h, r_h) in zip(dev[0], host[0]):
        assert L_d == L_h
        assert abs(e_d - e_h) <= 1e-12 * abs(e_h) and abs(r_d - r_h) <= 1e-9 * abs(r_h)
    np.testing.assert_allclose(dev[1], host[1], rtol=1e-10, atol=1e-14)
    assert dev[2] == host[2]
    for (e_d, _, _), e_ref in zip(dev[0], g['elbo']):
        assert abs(e_d - e_ref) < 1e-9 * abs(e_ref)     # ... and the reference's own trajectory
