"""Host driver logic (line search, L dynamics, M-step, error scaling, convergence) of the
product's MultiPopVI, exercised on CPU through the oracle-backed test engine, against the
trajectories recorded from the reference.  Also covers the world_size-2 sharded path (gloo)."""
import os
import sys

import numpy as np
import pytest

from helpers import golden, product_vi_from_traj, check_trajectory, engine_class, TRAJ_NAMES
from oracle_engine import OracleEngine


@pytest.mark.parametrize('name', TRAJ_NAMES)
def test_driver_trajectory(name):
    g = golden('traj_%s.npz' % name)
    vi, ld = product_vi_from_traj(g, engine_factory=OracleEngine)
    for p in range(int(g['P'])):                    # SNP -> block assignment is bit exact
        assert np.array_equal(ld[p].perm, g['perm'])
        assert np.array_equal(ld[p].missing, g['missing'])
        for b, blk in enumerate(ld[p].matrices):
            assert blk.s.shape[0] == int(g['rank_%d_%d' % (p, b)])
    check_trajectory(vi, g)


@pytest.mark.parametrize('name', ['p1_dense', 'p2_scale_se'])
def test_driver_optimize(name):
    g = golden('traj_%s.npz' % name)
    cap = {'p1_dense': 40, 'p2_scale_se': 30}[name]
    vi, _ = product_vi_from_traj(g, num_its=cap, engine_factory=OracleEngine)
    np.random.seed(int(g['seed']))
    params = vi.optimize()
    assert vi.num_its_run == int(g['opt_num_its'])
    np.testing.assert_allclose(vi.real_posterior_mean(params), g['opt_post_mean'], rtol=1e-6,
                               atol=1e-12)
    np.testing.assert_allclose(vi.error_scaling, g['opt_error_scaling'], rtol=1e-8)
    np.testing.assert_allclose(params[0], g['opt_vi_mu'], rtol=1e-6, atol=1e-12)


def test_resume_from_checkpoint():
    """optimize(loaded) continues from saved params and skips the 10-iteration guard
    (variational_inference.py:345-352, 381)."""
    g = golden('traj_p2_scale_se.npz')
    vi, _ = product_vi_from_traj(g, num_its=30, engine_factory=OracleEngine)
    np.random.seed(int(g['seed']))
    params = vi.optimize()
    dump = vi.create_dump_dict(params)
    dump = {k: np.array(v) for k, v in dump.items()}
    vi2, _ = product_vi_from_traj(g, num_its=30, engine_factory=OracleEngine)
    params2 = vi2.optimize(dump)
    # the same resume through the oracle restatement of the reference loop
    from helpers import oracle_from_traj
    ovi, _ = oracle_from_traj(g, num_its=30)
    oparams = ovi.optimize(dump)
    assert vi2.num_its_run == ovi.num_its_run < 10
    np.testing.assert_allclose(vi2.real_posterior_mean(params2), ovi.real_posterior_mean(*oparams),
                               rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(vi2.error_scaling, ovi.error_scaling, rtol=1e-9)


def test_validation_errors():
    from vilma_amd.variational_inference import MultiPopVI
    g = golden('traj_p1_scaled.npz')
    vi, ld = product_vi_from_traj(g, engine_factory=OracleEngine)
    kw = dict(marginal_effects=g['betahat'], std_errs=g['se'], ld_mats=ld,
              mixture_covs=list(g['covs']), annotations=g['annotations'], gwas_N=g['gwas_N'],
              init_hg=g['init_hg'], num_its=3)
    for drop in ('init_hg', 'gwas_N', 'num_its', 'annotations', 'std_errs'):
        bad = dict(kw)
        bad[drop] = None
        with pytest.raises(ValueError), engine_class(OracleEngine):
            MultiPopVI(**bad)
    bad = dict(kw); bad['marginal_effects'] = np.where(np.arange(g['betahat'].size).reshape(g['betahat'].shape) == 3, np.nan, g['betahat'])
    with pytest.raises(ValueError), engine_class(OracleEngine):
        MultiPopVI(**bad)
    bad = dict(kw); bad['mixture_covs'] = [-np.eye(1)] * 3
    with pytest.raises(ValueError), engine_class(OracleEngine):
        MultiPopVI(**bad)
    bad = dict(kw); bad['annotations'] = np.zeros_like(g['annotations'])
    with pytest.raises(ValueError), engine_class(OracleEngine):
        MultiPopVI(**bad)
    bad = dict(kw); bad['ld_mats'] = ld + ld
    with pytest.raises(ValueError), engine_class(OracleEngine):
        MultiPopVI(**bad)


def _supplied_delta_checks(vi, ovi, seed):
    """elbo(params) / real_posterior_*(params) evaluate the vi_delta they are handed, as the
    reference does (variational_inference.py:412-417, 740-751; its test_MultiPopVI_elbo,
    tests/test.py:1517-1525, calls elbo on a hand-made tuple), while the fit itself continues
    from the fixed point."""
    np.random.seed(seed)
    mu, delta, hyper = ovi._initialize()
    np.random.seed(seed)
    params = vi._initialize()
    # a device-resident tuple needs no check at all
    assert vi.elbo(params) == vi._objective and vi._given is None
    rng = np.random.default_rng(3)
    other = rng.dirichlet(np.ones(delta.shape[1]), size=delta.shape[0])    # not the fixed point
    want, got = ovi.elbo((mu, other, hyper)), vi.elbo((mu, other, hyper))
    assert vi._given is not None and vi._given['max_dev'] > 1e-3
    assert abs(want - got) <= 1e-9 * abs(want)
    # the reference's identity at that point: elbo = loglik - beta_KL - annotation_KL
    assert np.isclose(want, ovi._log_likelihood((mu, other, hyper)) - ovi._beta_KL(mu, other, hyper))
    np.testing.assert_allclose(vi.real_posterior_mean(mu, other, hyper),
                               ovi.real_posterior_mean(mu, other, hyper), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(vi.real_posterior_variance(mu, other, hyper),
                               ovi.real_posterior_variance(mu, other, hyper), rtol=1e-8,
                               atol=1e-16)
    # the fixed point itself is recognised (no second evaluation kept) and gives the usual ELBO
    assert abs(vi.elbo((mu, delta, hyper)) - ovi.elbo((mu, delta, hyper))) <= 1e-9 * abs(want)
    assert vi._given is None
    # a sweep started from the foreign vi_delta continues from the fixed point, with a warning
    L = np.ones(5)
    p2, L, e2, _ = vi._optimize_step((mu, other, hyper), L, got, 2., None)
    assert vi._given is None and np.isfinite(e2)


def test_elbo_honours_a_supplied_vi_delta():
    from helpers import oracle_from_traj
    g = golden('traj_p2_scale_se.npz')
    vi, _ = product_vi_from_traj(g, engine_factory=OracleEngine)
    ovi, _ = oracle_from_traj(g)
    _supplied_delta_checks(vi, ovi, int(g['seed']))


def test_every_rank_detects_an_empty_shard():
    """Fewer LD components than ranks: the deterministic plan is the same everywhere, so every
    rank raises before the first collective (nobody is left waiting in an all-reduce)."""
    from vilma_amd.variational_inference import MultiPopVI

    class FakeComm:
        active, backend, group = False, 'gloo', None

        def __init__(self, rank, world):
            self.rank, self.world = rank, world
    g = golden('traj_p1_dense.npz')
    _, ld = product_vi_from_traj(g, engine_factory=OracleEngine)
    n_comp = len(ld[0].matrices) + len(g['missing'])
    kw = dict(marginal_effects=g['betahat'], std_errs=g['se'], ld_mats=ld,
              mixture_covs=list(g['covs']), annotations=g['annotations'], gwas_N=g['gwas_N'],
              init_hg=g['init_hg'], num_its=3)
    for rank in (0, n_comp):                  # a rank that has SNPs and one that has none
        with pytest.raises(ValueError, match='no SNPs'), engine_class(OracleEngine):
            MultiPopVI(_comm=FakeComm(rank, n_comp + 1), **kw)


def _rank_main(rank, world, port, name, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                            world_size=world)
    try:
        g = golden('traj_%s.npz' % name)
        vi, _ = product_vi_from_traj(g, engine_factory=OracleEngine)
        assert vi.comm.world == world and 0 < len(vi._snps) < int(g['N'])
        check_trajectory(vi, g)
        q.put((rank, 'ok', len(vi._snps)))
    except BaseException as exc:     # noqa: BLE001 - report to the parent
        import traceback
        q.put((rank, 'fail', traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['p2_lowrank', 'p2_scale_se'])
def test_two_ranks_gloo(name):
    """LD blocks sharded over 2 processes; sums all-reduced over gloo; both ranks take the same
    accept/reject branches and reproduce the single-process (reference) trajectory."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in results:
        assert status == 'ok', 'rank %d failed:\n%s' % (rank, info)
    g = golden('traj_%s.npz' % name)
    assert sum(info for _, _, info in results) == int(g['N'])


def _eight_rank_problem():
    """2 cohorts, 12 LD blocks on the same partition (so 12 independent components to share out),
    a few LD-missing SNPs, seeded."""
    rng = np.random.default_rng(11)
    sizes = [14, 9, 22, 17, 11, 25, 8, 19, 13, 16, 10, 21]
    P, N, M = 2, int(np.sum(sizes)) + 7, 9
    n_ld = int(np.sum(sizes))
    order = rng.permutation(N)
    perm = np.concatenate([order[:n_ld], np.sort(order[n_ld:])]).astype(np.int64)
    missing = np.sort(order[n_ld:]).astype(np.int64)
    i = [np.arange(n) for n in sizes]
    blocks = [[rng.uniform(0.2, 0.9) ** np.abs(ix[:, None] - ix[None, :]) for ix in i] for _ in range(P)]
    se = rng.uniform(0.01, 0.05, size=(P, N))
    betahat = rng.normal(size=(P, N)) * se * 1.5
    betahat[:, missing] = 0.0
    se[:, missing] = 1.0
    covs = [v * (0.6 * np.eye(P) + 0.4 * np.ones((P, P))) for v in np.geomspace(1e-7, 1e-2, M)]
    return dict(P=P, N=N, M=M, perm=perm, missing=missing, blocks=blocks, se=se, betahat=betahat,
                covs=covs, ann=np.ones((N, 1)))


def _eight_main(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                            world_size=world)
    try:
        from oracle.ldop import EigenBlock, BlockDiagonalLD
        from oracle.vi import MultiPopVIOracle
        from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
        from vilma_amd.variational_inference import MultiPopVI
        pr = _eight_rank_problem()
        common = dict(marginal_effects=pr['betahat'], std_errs=pr['se'], mixture_covs=pr['covs'],
                      annotations=pr['ann'], checkpoint=False, gwas_N=np.full(pr['P'], 5e4),
                      init_hg=np.full(pr['P'], 0.2), num_its=6)
        ovi = MultiPopVIOracle(ld_mats=[BlockDiagonalLD([EigenBlock(X, 1.0) for X in pr['blocks'][p]],
                                                        perm=pr['perm'], missing=pr['missing'])
                                        for p in range(pr['P'])], **common)
        from helpers import engine_class as _engine_class
        with _engine_class(OracleEngine):
            vi = MultiPopVI(ld_mats=[BlockDiagonalMatrix([LowRankMatrix(X, 1.0) for X in pr['blocks'][p]],
                                                         perm=pr['perm'], missing=pr['missing'])
                                     for p in range(pr['P'])], **common)
        assert vi.comm.world == world and 0 < len(vi._snps) < pr['N']
        np.random.seed(7)
        op = ovi._initialize()
        np.random.seed(7)
        pp = vi._initialize()
        oe, pe = ovi.elbo(op), vi.elbo(pp)
        assert abs(oe - pe) <= 1e-9 * abs(oe)
        oL, pL, ored, pred = np.ones(5), np.ones(5), None, None
        for it in range(6):
            op, oL, oe, ored = ovi._optimize_step(op, oL, oe, 2., ored)
            pp, pL, pe, pred = vi._optimize_step(pp, pL, pe, 2., pred)
            assert abs(oe - pe) <= 1e-9 * abs(oe), (it, oe, pe)
            assert np.array_equal(oL, pL), (it, oL, pL)
        np.testing.assert_allclose(vi.real_posterior_mean(pp), ovi.real_posterior_mean(*op),
                                   rtol=1e-7, atol=1e-12)
        q.put((rank, 'ok', len(vi._snps)))
    except BaseException:     # noqa: BLE001 - report to the parent
        import traceback
        q.put((rank, 'fail', traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_gloo():
    """The driver's largest run: 8 ranks.  Shard plan (every rank gets blocks), the per-decision
    all-reduce protocol and the optimize loop with eight processes over gloo reproduce the
    unsharded oracle sweep by sweep.  (On a GPU box at most six processes may hold the card, so
    the eight-rank rehearsal runs here on the CPU engine; the GPU rehearsals stop at four.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_eight_main, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in results:
        assert status == 'ok', 'rank %d failed:\n%s' % (rank, info)
    assert sum(info for _, _, info in results) == _eight_rank_problem()['N']


def _gather_main(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                            world_size=world)
    try:
        from vilma_amd.sharding import Comm
        comm = Comm()
        comm.GATHER_CHUNK_BYTES = 4096          # force the slice-by-slice path ([7,2,n]: 3 slices)
        N = 1001
        rng = np.random.default_rng(0)
        owner = rng.integers(0, world, N)
        full = rng.normal(size=(7, 2, N))
        mine = np.flatnonzero(owner == rank)
        for arr in (full, full[0], full[0, 0]):          # 3-, 2- and 1-dimensional
            got = comm.gather_snps(arr[..., mine], mine, N)
            assert np.array_equal(got, arr)
        q.put((rank, 'ok', len(mine)))
    except BaseException:     # noqa: BLE001 - report to the parent
        import traceback
        q.put((rank, 'fail', traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gather_snps_in_slices_two_ranks():
    """Comm.gather_snps assembles ragged, interleaved shards in global SNP order on every rank,
    also when the array goes through in slices of its first axis."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gather_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in results:
        assert status == 'ok', 'rank %d failed:\n%s' % (rank, info)
    assert sum(info for _, _, info in results) == 1001


def test_shard_plan_is_closed_and_balanced():
    from vilma_amd.sharding import plan_shards, local_ld
    g = golden('traj_p2_lowrank.npz')
    vi, ld = product_vi_from_traj(g, engine_factory=OracleEngine)
    N = int(g['N'])
    for world in (1, 2, 3, 5):
        plan = plan_shards(ld, N, world, per_snp_cost=100.0)
        allsnps = np.concatenate([s['snps'] for s in plan])
        assert np.array_equal(np.sort(allsnps), np.arange(N))
        for p in range(2):
            allb = np.concatenate([s['blocks'][p] for s in plan])
            assert np.array_equal(np.sort(allb), np.arange(len(ld[p].matrices)))
        for s in plan:
            for p in range(2):
                mats, perm, n_ld = local_ld(ld[p], s['snps'], s['blocks'][p], N)
                assert sorted(perm.tolist()) == list(range(len(s['snps'])))
                assert n_ld == sum(m.shape[0] for m in mats)


def _agree_main(rank, world, port, bad_rank, mode, q):
    if rank == bad_rank and mode == 'noload':
        os.environ['VILMA_RCCL_LIB'] = '/nonexistent/librccl.so'      # before the library loads
    import ctypes as C
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                            world_size=world)
    try:
        from vilma_amd import _lib
        from vilma_amd.sharding import Comm, agree_on_rccl
        lib = _lib.load()
        comm = Comm()
        calls = []

        def make_id():          # the library's own entry point, as HipEngine.bind_comm calls it
            ident = C.create_string_buffer(128)
            return ident.raw if lib.vilma_comm_unique_id(ident) == 0 else None

        def init_rccl(raw):     # ncclCommInitRank needs a GPU: stand-in that fails where asked
            calls.append(len(raw))
            if mode == 'initfail' and rank == bad_rank:
                raise RuntimeError('ncclCommInitRank: simulated failure')

        ok, err, failed = agree_on_rccl(comm, make_id, init_rccl)
        # every rank then runs its sweeps through the callback collective: the sharded fit is the
        # reference trajectory whatever the agreement said
        g = golden('traj_p2_scale_se.npz')
        vi, _ = product_vi_from_traj(g, engine_factory=OracleEngine)
        assert vi.comm.world == world
        check_trajectory(vi, g)
        q.put((rank, 'ok', (ok, err, failed, len(calls))))
    except BaseException:     # noqa: BLE001 - report to the parent
        import traceback
        q.put((rank, 'fail', traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['noload', 'initfail'])
def test_rccl_setup_failure_on_one_of_four_ranks_switches_every_rank(mode):
    """One rank of four cannot load librccl (the library's real failure path, VILMA_RCCL_LIB) or
    fails in ncclCommInitRank: every rank learns it, nobody is left inside ncclCommInitRank, all
    four fall back to the callback collective together and the fit is unchanged."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (0 if mode == 'noload' else 1)
    procs = [ctx.Process(target=_agree_main, args=(r, 4, port, 2, mode, q)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in results:
        assert status == 'ok', 'rank %d failed:\n%s' % (rank, info)
    for rank, _, (ok, err, failed, n_calls) in results:
        assert ok is False and err
        if mode == 'noload':
            # known before anybody calls ncclCommInitRank
            assert n_calls == 0 and 'not loadable on 1 rank' in err
        else:
            assert n_calls == 1 and failed == 1
            assert ('simulated failure' in err) == (rank == 2)


def test_bench_line_of_four_ranks_carries_the_fields_a_scale_record_is_judged_on():
    """The driver's multi-GPU launch line, rehearsed on the CPU (tests/bench_rehearsal.py = bench.py's
    main() with the oracle-backed test engine over gloo, no GPU, `metric` says REHEARSAL): whatever
    N -- 8 included, the size of the node the driver measures on -- the ONE JSON line carries
    rccl_ranks == N, the collective, per-rank times and shard sizes and the slowest rank's roofline,
    has no CPU-baseline leg, and the whole launch finishes within a minute."""
    import json
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for n in (2, 4, 8):
        env = dict(os.environ, OMP_NUM_THREADS='1')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
               str(n), '--master-addr', '127.0.0.1', '--master-port',
               str(34100 + os.getpid() % 800 + n), os.path.join('tests', 'bench_rehearsal.py'),
               '--gpus', str(n), '--steps', '2', '--warmup', '1', '--workload',
               'tiny8' if n == 8 else 'tiny']
        t0 = time.perf_counter()
        out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        wall = time.perf_counter() - t0
        assert out.returncode == 0, out.stderr[-3000:]
        assert wall < 60.0, 'the %d-rank rehearsal took %.0f s' % (n, wall)
        lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, out.stdout
        d = json.loads(lines[0])
        assert d['metric'].startswith('REHEARSAL') and d['n_gpus'] == n and d['steps'] == 2
        assert d['rccl_ranks'] == n and d['collective']
        pr = d['per_rank']
        for key in ('ms_per_step', 'ld_algorithmic_bytes', 'snps', 'avg_launch_ms', 'launches',
                    'achieved_GBps'):
            assert len(pr[key]) == n, key
        assert min(pr['snps']) > 0 and (n == 8 or sum(pr['snps']) == 6316)
        assert d['ms_per_step_min_rank'] <= d['ms_per_step_max_rank']
        slow = d['roofline_slowest_rank']
        for key in ('rank', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms',
                    'launches', 'algorithmic_bytes_per_launch'):
            assert key in slow, key
        assert 0 <= slow['rank'] < n and slow['peak'] == 8000.0
        assert 'cpu_baseline' not in d and d['scaling'] == 'strong'
