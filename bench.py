#!/usr/bin/env python3
"""Headline benchmark: full VI sweeps/s of the `vilma fit` loop on synthetic data.

    python bench.py --gpus N --steps K --warmup W [--workload C3]

A "step" is one outer sweep of the fit loop (reference MultiPopVI._optimize_step + the
convergence statistics, variational_inference.py:361-389) over the whole synthetic problem.
Workload C3 (BASELINE.json configs[2], the one the metric is quoted on): 2 cohorts, 1,000,000
SNPs in 1700 LDetect-sized AR(1) LD blocks (+5% LD-missing SNPs), M=40 mixture components,
dense-rank LD held in fp64 (11.98 GB), inputs resident in HBM before the timed region.
With N>1 (launched by torch.distributed.run, one rank per GPU) the same global problem is
sharded by LD blocks (strong scaling); the 3P+2 sums are all-reduced over RCCL per evaluation.

Prints ONE JSON line (rank 0) with the sweep throughput, the roofline object of the dominant
kernel (ld_tile_kernel for dense LD -- ld_sym_kernel with VILMA_LD_TILE=0 --, ld_eig_fused_kernel
for eigen-form LD; timed with HIP events on its launch stream inside the library) and, at N=1, the CPU baseline: the oracle (a port following
the reference's operation schedule) timed on this host on a bounded sample of the same workload.

roofline.achieved is priced on the ALGORITHMIC bytes of the product as this build defines it:
a symmetric dense block needs its lower triangle once, 8 n(n+1)/2 bytes; an eigen-form block
needs U once, 8 n r.  SURVEY 8d's dense figure 8 n^2 (both triangles) is kept as
the secondary field `survey_8d_bytes_per_launch`; it is not a utilisation basis for a kernel
that legitimately reads half the matrix.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0   # measured copy ceiling on MI355X (same guide, HBM section)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50,
                    help='timed sweeps (SURVEY.md 8d quotes the metric over a run of 50)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='C3')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--ld-form', default='auto', choices=['auto', 'dense', 'eig'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--learn-scaling', action='store_true',
                    help='fit with --learn-scaling (scale_se: the error-scaling EM update of '
                         '_update_error_scaling, reference variational_inference.py:441-448, 472-486, '
                         'runs whenever a sweep gains less than EM_TOL); the config string says so')
    ap.add_argument('--cpu-frac', type=float, default=None,
                    help='fraction of the workload\'s LD blocks in the CPU-baseline sample '
                         '(default: 0.10 -- about 10 s of timed CPU sweeps at C3 after ~40 - 70 s of '
                         'untimed setup, eigendecompositions and the ridge start of the sample; '
                         '0.02 for C5, whose sweep is ~25x C3\'s on the CPU)')
    ap.add_argument('--cpu-sweeps', type=int, default=12,
                    help='sweeps of the CPU baseline (fewer if --cpu-budget runs out first)')
    ap.add_argument('--cpu-budget', type=float, default=150.0,
                    help='seconds the whole CPU-baseline leg may take (sample setup included): '
                         'the timed loop stops after the sweep during which 60 %% of the budget is '
                         'gone (at least 2 sweeps) and the in-run GPU-vs-CPU comparison of the '
                         'sample is skipped when less than a fifth of it is left')
    ap.add_argument('--prof-every', type=int, default=0,
                    help='bracket every k-th LD product with HIP events (roofline.avg_launch_ms); '
                         'default: every product on 1 GPU, every 8th on a sharded run, where the '
                         '~10 us of stream time an event pair costs is percents of a sweep')
    ap.add_argument('--emulate-shard', type=int, default=0,
                    help='diagnostic: time rank 0 of a K-way sharding alone on one GPU '
                         '(no collectives; NOT a valid bench line)')
    ap.add_argument('--emulate-rank', type=int, default=0,
                    help='with --emulate-shard K: which rank\'s shard to time (default 0)')
    return ap.parse_args(argv)


def cpu_baseline(workload, seed, block_frac, n_sweeps, budget_s=150.0, scale_se=False):
    """The oracle (oracle/vi.py: the reference's operation schedule -- two GEMVs per LD block in
    a serial block loop with threaded BLAS, 5-8 products per sweep, one pass per numerics
    function) on the first `block_frac` of the blocks of the same synthetic problem, on this
    host, with the per-SNP passes as compiled C / OpenMP loops over all cores (oracle/
    numerics_omp.c: what numba `prange` gives the reference).  Returns the JSON object."""
    from concurrent.futures import ThreadPoolExecutor
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS, ar1_numpy
    from oracle.ldop import EigenBlock, BlockDiagonalLD
    from oracle.vi import MultiPopVIOracle
    from oracle import native
    from threadpoolctl import threadpool_limits
    t_leg = time.perf_counter()
    cfg = dict(WORKLOADS[workload])
    lowrank = cfg.get('kind', 'ar1') != 'ar1'
    full = SyntheticShard(seed=seed, **cfg)
    n_blocks = max(1, min(len(full.sizes_all), int(round(block_frac * len(full.sizes_all)))))
    # (the eigen-form workloads C4 / C4f build the sample's factors (U, s) on this process's GPU,
    # as the timed run's loader did, and hand them to the CPU leg as host arrays: setup, not timed)
    device = None
    if lowrank:
        import torch
        device = torch.device('cuda', torch.cuda.current_device())
    sh = SyntheticShard(seed=seed, block_range=(0, n_blocks), **cfg).build(device)
    P = sh.P
    # every core this process may USE: the affinity mask, cut by the cgroup's CPU quota (a GPU box
    # hands out one GPU's share of the host) and by the 64 threads this image's OpenBLAS was built
    # for (it crashes beyond that); BLAS and OpenMP both get all of them, as numba's prange and
    # threaded BLAS do for the reference
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        txt = open('/sys/fs/cgroup/cpu.max').read().split()
        if txt[0] != 'max':
            quota = max(1, int(float(txt[0]) / float(txt[1])))
    except (OSError, ValueError, IndexError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = max(1, q // per)
        except (OSError, ValueError):
            pass
    threads = max(1, min(avail, quota or avail, 64))
    core_note = 'affinity mask %d, cgroup quota %s, OpenBLAS build limit 64' % (avail, quota or 'none')
    # setup (not timed): eigendecompose the sample's blocks, one LAPACK call per core
    if lowrank:
        ld = [BlockDiagonalLD([EigenBlock(u=U.cpu().numpy(), s=sv.cpu().numpy(), t=1.0)
                               for U, sv in sh._eig[p]], perm=sh.perm, missing=sh.missing)
              for p in range(P)]
    else:
        with threadpool_limits(limits=1), ThreadPoolExecutor(max_workers=threads) as pool:
            ld = [BlockDiagonalLD(list(pool.map(lambda b: EigenBlock(ar1_numpy(b.n, b.rho[p]), 1.0),
                                                sh.blocks)), perm=sh.perm, missing=sh.missing)
                  for p in range(P)]
    annotations = np.ones((sh.N, 1))
    vi = MultiPopVIOracle(marginal_effects=sh.betahat, std_errs=sh.se, ld_mats=ld,
                          annotations=annotations, mixture_covs=list(sh.covs), checkpoint=False,
                          checkpoint_freq=-1, scaled=False, scale_se=scale_se, gwas_N=sh.gwas_N,
                          init_hg=sh.init_hg, num_its=n_sweeps)
    cpu_elbos = []
    # the timed region is split into LD products and per-SNP passes (shares reported)
    ld_seconds = [0.0]
    for op in ld:
        def timed_dot(x, _dot=op.dot):
            t = time.perf_counter()
            y = _dot(x)
            ld_seconds[0] += time.perf_counter() - t
            return y
        op.dot = timed_dot
    native.enable(threads=threads)
    try:
        with threadpool_limits(limits=threads):
            np.random.seed(42)
            params = vi._initialize()
            elbo = vi.elbo(params)
            cpu_elbos.append(elbo)
            L, red = np.ones(5), None
            params, L, elbo, red = vi._optimize_step(params, L, elbo, 2., red)   # warm-up sweep
            cpu_elbos.append(elbo)
            ld_seconds[0] = 0.0
            setup_s = time.perf_counter() - t_leg
            t0 = time.perf_counter()
            asked, n_sweeps = n_sweeps, 0
            for _ in range(asked):
                params, L, elbo, red = vi._optimize_step(params, L, elbo, 2., red)
                cpu_elbos.append(elbo)
                n_sweeps += 1
                # bounded by time, not by count: a slow host must not cost the bench line
                if n_sweeps >= 2 and time.perf_counter() - t_leg > 0.6 * budget_s:
                    break
            dt = time.perf_counter() - t0
    finally:
        native.disable()
    t_ld = min(ld_seconds[0], dt)
    frac = sh.N / full.N_global
    parity = None
    try:
        if time.perf_counter() - t_leg > 0.8 * budget_s:
            raise TimeoutError('skipped: the CPU leg had used %.0f s of its %.0f s budget; the '
                               'same comparison is asserted by tests/test_gpu_fullsize.py'
                               % (time.perf_counter() - t_leg, budget_s))
        # "ELBO vs CPU": the same sample through the product class API on the GPU, same seed
        from vilma_amd.matrix_structures import LowRankMatrix, BlockDiagonalMatrix
        from vilma_amd.variational_inference import MultiPopVI
        gld = [BlockDiagonalMatrix([LowRankMatrix(u=b.u, s=b.s, v=b.v, D=np.zeros(b.u.shape[0]))
                                    for b in ld[p].blocks], perm=sh.perm, missing=sh.missing)
               for p in range(P)]
        gvi = MultiPopVI(marginal_effects=sh.betahat, std_errs=sh.se, ld_mats=gld,
                         annotations=annotations, mixture_covs=list(sh.covs), checkpoint=False,
                         scaled=False, scale_se=scale_se, gwas_N=sh.gwas_N, init_hg=sh.init_hg,
                         num_its=n_sweeps + 1)
        np.random.seed(42)
        gp = gvi._initialize()
        gelbo, gL, gred = gvi.elbo(gp), np.ones(5), None
        gpu_elbos = [gelbo]
        for _ in range(n_sweeps + 1):
            gp, gL, gelbo, gred = gvi._optimize_step(gp, gL, gelbo, 2., gred)
            gpu_elbos.append(gelbo)
        cm = vi.real_posterior_mean(*params)
        gm = gvi.real_posterior_mean(gp)
        dev = np.abs(np.array(gpu_elbos) - np.array(cpu_elbos)) / np.abs(np.array(cpu_elbos))
        parity = {'sweeps': n_sweeps + 1, 'elbo_max_rel_dev': float(dev.max()),
                  'post_mean_max_abs_dev': float(np.abs(gm - cm).max()),
                  'post_mean_max_rel_dev_where_gt_1e-6': float(
                      (np.abs(gm - cm) / np.maximum(np.abs(cm), 1e-300))[np.abs(cm) > 1e-6].max()),
                  'L_trajectory_equal': bool(np.array_equal(gL, L)),
                  'bar': 'ELBO and posterior means within 1e-5 relative'}
        gvi.engine.close()
    except Exception as exc:
        parity = {'error': repr(exc)}
    return {
        'value': (n_sweeps / dt) * frac, 'unit': 'sweeps/s', 'cores': int(threads),
        'cores_os_cpu_count': os.cpu_count(), 'cores_how': core_note, 'sweeps_timed': int(n_sweeps),
        'sweeps_asked': int(asked), 'budget_seconds': budget_s, 'setup_seconds': setup_s,
        'leg_seconds': time.perf_counter() - t_leg,
        'kind': 'port', 'parity_vs_cpu': parity,
        'ld_product_share_of_cpu_time': t_ld / dt,
        'sample_fraction_of_snps': frac,
        'sample': ('oracle with the reference\'s schedule (per sweep 5-8 LD products per cohort as '
                   'two BLAS GEMVs per block in a serial block loop, one pass per numerics '
                   'function; per-SNP passes as compiled C/OpenMP loops) on %d threads, on the '
                   'first %d of %d blocks (%d of %d SNPs = %.1f %%, all %d cohorts, M=%d): %d '
                   'sweeps in %.2f s after 1 warm-up, %.0f %% of it in LD products; value = '
                   'measured sample sweeps/s x SNP fraction (per-sweep cost is linear in SNPs '
                   'for a fixed block-size law)'
                   % (threads, n_blocks, len(full.sizes_all), sh.N, full.N_global, 100 * frac, P,
                      sh.M, n_sweeps, dt, 100 * t_ld / dt)),
    }


def kernels_sha16():
    import hashlib
    src = os.path.join(ROOT, 'vilma_amd', 'csrc', 'kernels.hip')
    return hashlib.sha256(open(src, 'rb').read()).hexdigest()[:16]


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the driver's own launch line
    as a CHILD process -- before this process has touched the GPU -- relay its output and exit
    with its code, so the documented command can never silently measure one GPU."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main(argv=None, engine_factory=None):
    """`engine_factory` is for tests/bench_rehearsal.py only: a CPU test engine with HipEngine's
    interface, to rehearse N ranks' protocol and the fields of the JSON line without a GPU (over
    gloo).  Such a line says REHEARSAL in `metric` and measures nothing.  No environment variable
    or flag of this program selects an engine: run as `python bench.py`, it is always HipEngine."""
    args = parse(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(args)
    import torch
    import torch.distributed as dist
    from vilma_amd.synthetic import SyntheticShard, WORKLOADS
    from vilma_amd.engine import HipEngine
    from vilma_amd.sharding import Comm
    from vilma_amd.variational_inference import SweepDriver

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    # rehearsal knobs (tests only): several ranks on one GPU over gloo
    rehearsal = engine_factory is not None
    engine_spec = getattr(engine_factory, '__name__', 'test engine') if rehearsal else None
    if os.environ.get('VILMA_BENCH_SAME_DEVICE') == '1':
        local_rank = 0
    backend = os.environ.get('VILMA_BENCH_BACKEND', 'gloo' if rehearsal else 'nccl')
    device = None
    if not rehearsal:
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)

    def sync():
        if not rehearsal:
            torch.cuda.synchronize()
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    force = os.environ.get('VILMA_BENCH_FORCE_RCCL') == '1'
    if force and world == 1:
        # diagnostic: a one-rank RCCL group so every decision pays a real all-reduce launch
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29621')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
    comm = Comm(force=force)
    # N = 1 under the launcher (WORLD_SIZE=1) is the plain run: no process group, no collective

    cfg = dict(WORKLOADS[args.workload])
    t_setup = time.perf_counter()
    if args.emulate_shard > 1:
        shard = SyntheticShard(seed=args.seed, rank=args.emulate_rank % args.emulate_shard,
                               world=args.emulate_shard, **cfg).build(device)
    else:
        shard = SyntheticShard(seed=args.seed, rank=rank, world=world, **cfg).build(device)
    P, M = shard.P, shard.M
    g = comm.allreduce_np(np.concatenate([shard.chi_local, shard.rank_local, shard.inv_se2_local]))
    chi, ranks, inv_se2 = g[:P], g[P:2 * P], g[2 * P:]
    shard.finish_init(inv_se2)

    if rehearsal:
        engine = engine_factory(P, shard.N, M, 1)
    else:
        engine = HipEngine(P, shard.N, M, 1)
    engine.set_snp_data(shard.adj, shard.se, shard.sld, shard.scalings, shard.annot)
    prec = np.linalg.inv(shard.covs)
    log_det = np.linalg.slogdet(shard.covs)[1]
    engine.set_mixture(prec, log_det)
    for p in range(P):
        if rehearsal:
            engine.load_ld(p, list(shard.ld_blocks_numpy(p)), shard.perm, shard.n_ld)
        elif shard.kind == 'lowrank':
            engine.load_ld(p, shard.ld_blocks_torch(p, device, args.ld_form), shard.perm,
                           shard.n_ld, specs=shard.block_specs(args.ld_form, p))
        else:
            engine.load_ld(p, shard.ld_blocks_torch(p, device), shard.perm, shard.n_ld,
                           specs=shard.block_specs())
    sync()

    driver = SweepDriver()
    driver._setup_driver(engine, comm, P, M, 1, chi, ranks, [shard.N_global], log_det,
                         scale_se=args.learn_scaling, num_its=args.steps + args.warmup)
    # one all-reduce through the sweep's own collective, on its stream, before anything is timed:
    # the communicator is connected and every rank is in it
    rccl_ranks = engine.comm_check()
    if comm.active and rccl_ranks != world:
        raise SystemExit('the collective reached %d rank(s), not %d' % (rccl_ranks, world))
    driver.initialize_from(shard.fake_mu)
    setup_s = time.perf_counter() - t_setup
    elbo0 = driver._objective

    # lookahead: the driver may queue the next sweep's M-step stage ahead of its line-search
    # decision (taken on the device).  The last warm-up sweep and the last timed sweep do not
    # look ahead, so exactly the K timed sweeps' work runs inside the timed region.
    state = None
    for w in range(args.warmup):
        state, _ = driver.sweep(state, lookahead=w + 1 < args.warmup)
    if state is None:
        engine.snapshot_mean()
        state = {'L': np.ones(5), 'elbo': driver._objective, 'running': None}

    # an event pair costs ~10 us of stream time: every 3rd product on one GPU (<1 % of a C3 sweep;
    # an odd stride, so one- and two-right-hand-side launches are both sampled), every 9th on a shard
    # (with --learn-scaling a group is trial + evaluation + re-evaluation: a stride of 3 would bracket
    # the same phase every time)
    prof_every = args.prof_every or ((4 if args.learn_scaling else 3)
                                     if (world == 1 and args.emulate_shard <= 1) else 9)
    engine.prof_enable(True, every=prof_every)
    engine.prof_read(reset=True)
    ev0, tr0, ah0 = driver.n_evaluations, driver.n_trials, driver.n_stages_ahead
    pr0 = driver.n_products
    sk0 = driver.n_stages_skipped
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    elbos = []
    tau_start = np.array(np.atleast_1d(driver.error_scaling), dtype=np.float64)
    tau_prev, tau_updates = tau_start, 0
    for k in range(args.steps):
        state, _ = driver.sweep(state, lookahead=k + 1 < args.steps)
        elbos.append(state['elbo'])
        if args.learn_scaling:      # (a host-side comparison of what the sweep reported: no sync)
            tau_now = np.array(np.atleast_1d(driver.error_scaling), dtype=np.float64)
            tau_updates += int(np.any(tau_now != tau_prev))
            tau_prev = tau_now
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    prof = engine.prof_read(reset=True)
    engine.prof_enable(False)
    # the yardstick for this process: a bare read of the very same LD store (where an allocation
    # lands in HBM moves a box's streaming rate by several percent from process to process)
    try:
        store_ms, store_bytes = engine.stream_store(5)
    except Exception:
        store_ms, store_bytes = None, 0
    # the dominant kernel = the LD-streaming kernel with the most accumulated time; ld_sym_kernel
    # launches with one and with two right-hand sides (a two-step beta trial: one pass over the
    # store, two products) are bracketed separately and pooled for the roofline: the algorithmic
    # bytes of a launch are the store once either way
    sym1, sym2 = prof['ld_sym_kernel'], prof['ld_sym_kernel_two_rhs']
    # (the library's bracket kinds keep round 1's names; the kernel behind the dense product is
    # ld_tile_kernel since round 5 -- ld_sym_kernel with VILMA_LD_TILE=0)
    tile_rows, tile_slabs, tile_items = engine.ld_tile() if hasattr(engine, 'ld_tile') else (0, 0, 0)
    sym_name = 'ld_tile_kernel' if tile_rows > 0 else 'ld_sym_kernel'
    pooled = {sym_name: (sym1[0] + sym2[0], sym1[1] + sym2[1]),
              'ld_eig_fused_kernel': prof['ld_eig_fused_kernel']}
    dom = max(pooled, key=lambda k: pooled[k][0])
    kernel_ms, launches = pooled[dom]
    # (an eigen-form product = the fused launches, one per block-height class, and their combine:
    # the library brackets them together, so a bracket is a product for either kernel)
    if world > 1:
        elapsed = float(comm.allreduce_np(np.array([elapsed]), op='max')[0])

    n_eval = driver.n_evaluations - ev0
    n_prod = driver.n_products - pr0
    # algorithmic bytes of ONE product on this rank (stated in DESIGN.md section 5): a dense
    # symmetric block needs its lower triangle once, an eigen-form block its U once
    alg_launch = 8.0 * sum(n * (n + 1) / 2 if form == 'dense' else n * r
                           for p in range(P)
                           for form, n, r in (shard.block_specs(args.ld_form, p)
                                              if shard.kind == 'lowrank' else shard.block_specs()))
    survey_launch = float(shard.ld_bytes)                # 8 n^2 dense, 8 n r eigen
    specs_all = [sp for p in range(P) for sp in (shard.block_specs(args.ld_form, p)
                                                  if shard.kind == 'lowrank' else shard.block_specs())]
    alg_dense = 8.0 * sum(n * (n + 1) / 2 for form, n, r in specs_all if form == 'dense')
    alg_eig = 8.0 * sum(n * r for form, n, r in specs_all if form != 'dense')
    avg_ms = kernel_ms / max(launches, 1)
    achieved = alg_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
    state_bytes = 8.0 * shard.N * (2 * M * P)            # per-SNP pass: read mu, write mu'

    # Every hot kernel of the sweep with its own algorithmic bytes and fraction of the HBM peak
    # (an LD launch reads its store once whatever the number of right-hand
    # sides; a per-SNP pass reads vi_mu [M][P][N] once and writes one array per candidate it stores;
    # the sums pass behind a lazy trial reads it and writes the accepted candidate).  All kinds are
    # bracketed on the same sampling tick, so launches_k / sampled products = launches of kind k
    # per LD product, and the sweep's algorithmic bytes follow from the exact product count.
    mp_bytes = 8.0 * shard.N * M * P
    # the form the queued sweeps held the state in (vilma_prof_state_form): 2 = mu_k = Sig_k c, what a fit
    # started by _initialize is and stays -- the lazy passes then read NO vi_mu array (their loads hit
    # one L2-resident tile): they are priced on the [P][N] vectors they do move and are ALU-bound
    state_form = engine.state_form() if hasattr(engine, 'state_form') else 0
    vec = 8.0 * shard.N * P
    lazy_bytes = {'snp_pass_eval': 11 * vec, 'snp_pass_trial_lazy': 12 * vec, 'snp_pass_trial2_lazy': 17 * vec,
                  'sums_pass': 2 * vec} if state_form == 2 else {}
    kinds = [(sym_name + '<1>', 'ld_sym_kernel', alg_dense),
             (sym_name + '<2>', 'ld_sym_kernel_two_rhs', alg_dense),
             ('ld_eig_fused_kernel', 'ld_eig_fused_kernel', alg_eig),
             ('snp_pass_eval', 'snp_pass_eval', mp_bytes),
             ('snp_pass_trial', 'snp_pass_trial', 2 * mp_bytes),
             ('snp_pass_trial2', 'snp_pass_trial2', 3 * mp_bytes),
             ('snp_pass_trial_lazy', 'snp_pass_trial_lazy', mp_bytes),
             ('snp_pass_trial2_lazy', 'snp_pass_trial2_lazy', mp_bytes),
             # (with --learn-scaling the pass also WRITES the state in the sweep after a tau update --
             # the library cannot tell when it queues it: bytes and fraction are then a lower bound)
             ('delta_kernel (sums pass%s)' % ('; storing the state too in sweeps behind a tau update'
                                              if args.learn_scaling else ''), 'sums_pass', mp_bytes),
             ('delta_kernel<MAT> (re-derive + store + sums)', 'sums_pass_store', 2 * mp_bytes)]
    kernel_rows = []
    for name, key, nbytes in kinds:
        ms_k, n_k = prof.get(key, (0.0, 0))
        if not n_k:
            continue
        base_free = key in lazy_bytes
        if base_free:
            nbytes = lazy_bytes[key]
        gbps = nbytes / (ms_k / n_k * 1e-3) / 1e9
        kernel_rows.append({'name': name, 'algorithmic_bytes': nbytes, 'avg_ms': ms_k / n_k,
                            'launches': int(n_k), 'GBps': gbps, 'frac': gbps / HBM_PEAK_GBS})
        if base_free:
            kernel_rows[-1]['bound'] = 'alu (the state is mu_k = Sig_k c: no vi_mu array is read; bytes = the [P][N] vectors moved)'
    sampled_products = sum(prof[k][1] for k in ('ld_sym_kernel', 'ld_sym_kernel_two_rhs')) or \
        prof['ld_eig_fused_kernel'][1]
    per_product = (sum(r['algorithmic_bytes'] * r['launches'] for r in kernel_rows) / sampled_products
                   if sampled_products else 0.0)
    sweep_bytes = per_product * n_prod / max(args.steps, 1)

    # HBM traffic of the dominant kernel: NOT measured in this run (PMC counters need their own
    # rocprofv3 passes); replayed from the committed PMC summary when it was taken on this very
    # kernel source and workload, and labelled as such
    traffic, traffic_source = None, 'none'
    tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    if os.path.exists(tpath) and world == 1 and args.emulate_shard <= 1:
        try:
            rec = json.load(open(tpath))
            same_build = rec.get('kernels_hip_sha16') == kernels_sha16()
            if rec.get('workload') == args.workload and rec.get('ld_form', 'auto') == args.ld_form \
                    and same_build and rec.get(dom + '_bytes_per_launch'):
                traffic = rec[dom + '_bytes_per_launch']
                traffic_source = 'replayed from %s (rocprofv3 --pmc passes of this kernel source ' \
                                 'on another box; not measured in this run)' % rec.get('source')
            elif not same_build:
                traffic_source = 'none: the PMC summary on file was taken on another kernel build'
        except Exception:
            traffic = None

    out = {
        'metric': ('REHEARSAL on a CPU test engine (%s): not a measurement' % engine_spec if rehearsal else
                   'DIAGNOSTIC shard emulation' if args.emulate_shard > 1 else
                   'full VI sweeps/sec (1M SNPs, 2 cohorts)' if args.workload == 'C3'
                   else 'full VI sweeps/sec'),
        'value': args.steps / elapsed, 'unit': 'sweeps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': '%s: %d cohorts, %d SNPs (%d in %d %s LD blocks + %d LD-missing), '
                        'M=%d mixture components%s, fp64 LD %.2f GB algorithmic (%s), A=1, %s'
                        % (args.workload, P, shard.N_global, shard.n_ld_global,
                                             len(shard.sizes_all),
                                             'AR(1)' if shard.kind == 'ar1' else ('factor-model, --ldthresh 0.8 (kept rank %.3f n)' % (sum(float(r.sum()) for r in shard.ranks_by_cohort) / (P * max(1.0, float(shard.sizes.sum())))) if shard.spectrum == 'factor' else 'eigen-form (rank %.2f n)' % shard.rank_frac),
                                             shard.N_global - shard.n_ld_global, M,
                                             (' (the grid `vilma fit` builds by default: _make_simple at -K %d)' % cfg['K']) if cfg.get('mixture') == 'make_simple' else '',
                                             (1e-9 * shard.ld_bytes if world == 1 and args.emulate_shard <= 1 else 8e-9 * P * float((shard.sizes_all.astype(np.float64) * shard.ranks_all).sum())),
                                             '8 n^2 as full matrices; the symmetric kernel needs the lower triangle, half of it' if shard.kind == 'ar1' else '8 n r, U counted once',
                                             'WITH --learn-scaling (error_scaling updated by EM)' if args.learn_scaling else 'no --learn-scaling'),
            'sharding': 'LD blocks over %d GPU(s), contiguous runs balanced by bytes' % world,
            'state_form': {0: 'stored vi_mu', 1: '(stored vi_mu, a, c)', 2: 'mu_k = Sig_k c (a = 0: no vi_mu array read by the sweeps)'}[state_form],
            'ld_work_items': ('%d per launch: tiles of %d rows x %d slabs of 128 columns (ld_tile_kernel)'
                              % (tile_items, tile_rows, tile_slabs)) if tile_rows > 0 else
                             ('%d per launch: one slab chunk each (ld_sym_kernel)' % tile_items),
            'points_evaluated_per_sweep': n_eval / args.steps,
            # a beta trial evaluates the step the line search tries now and the one it would try
            # next in ONE pass over the LD store: fewer passes than points
            'ld_products_per_sweep': n_prod / args.steps,
            'beta_trials_per_sweep': (driver.n_trials - tr0) / args.steps,
            'sweeps_queued_ahead_of_their_decision': driver.n_stages_ahead - ah0,
            # each skipped stage = 2 LD launches (and ~10 others) that exit at once: they show up
            # in a rocprofv3 --stats average as ~6 us launches, not in roofline.avg_launch_ms
            'stages_queued_ahead_then_skipped': driver.n_stages_skipped - sk0,
            'elbo_start': elbo0, 'elbo_end': elbos[-1] if elbos else elbo0,
            'error_scaling_end': [float(t) for t in np.atleast_1d(driver.error_scaling)],
            # --learn-scaling: the error scaling at the start of the timed region and in how many of
            # the timed sweeps the EM update moved it (it only acts once a sweep gains < EM_TOL:
            # ~158 sweeps into C3, so `--warmup 160` puts it inside the timed region)
            'error_scaling_start': [float(t) for t in tau_start],
            'sweeps_that_updated_error_scaling': tau_updates if args.learn_scaling else None,
            'setup_seconds': setup_s,
        },
        'roofline': {
            'bound': 'hbm', 'kernel': dom, 'achieved': achieved,
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'frac_of_achievable': achieved / HBM_ACHIEVABLE_GBS,
            'achievable_GBps': HBM_ACHIEVABLE_GBS,
            'traffic': traffic, 'traffic_source': traffic_source,
            # a read-only streaming kernel over the same store, same process, after the timed
            # region (vilma_prof_stream_store): what these bytes stream at with no arithmetic
            'store_stream': ({'ms': store_ms, 'bytes': store_bytes,
                              'GBps': store_bytes / (store_ms * 1e-3) / 1e9} if store_ms else None),
            'kernel_GBps_on_bytes_moved': ((traffic or alg_launch) / (avg_ms * 1e-3) / 1e9
                                           if launches else None),
            'bytes_moved_basis': 'PMC traffic' if traffic else 'algorithmic bytes (no PMC figure for this build)',
            'frac_of_store_stream': (((traffic or alg_launch) / (avg_ms * 1e-3))
                                     / (store_bytes / (store_ms * 1e-3))
                                     if launches and store_ms else None),
            'algorithmic_bytes_per_launch': alg_launch,
            'basis': 'symmetric dense block: lower triangle once, 8 n(n+1)/2 B; eigen-form '
                     'block: U once, 8 n r B; summed over this rank\'s blocks and cohorts',
            'survey_8d_bytes_per_launch': survey_launch,
            'avg_launch_ms': avg_ms, 'launches': int(launches), 'bracketed_every': prof_every,
            'stored_bytes_per_launch': float(engine.ld_bytes()[1]),
            'other_ld_kernels_ms': {k: v[0] / max(v[1], 1) for k, v in pooled.items() if k != dom and v[1]},
            # ld_sym_kernel by number of right-hand sides per pass (two = a two-step beta trial)
            'avg_launch_ms_one_rhs': sym1[0] / sym1[1] if sym1[1] else None,
            'avg_launch_ms_two_rhs': sym2[0] / sym2[1] if sym2[1] else None,
            'launches_two_rhs': int(sym2[1]),
            'sweep_algorithmic_GBps': (n_prod * (alg_launch + state_bytes)) / elapsed / 1e9,
            # every hot kernel on its own algorithmic bytes; the whole sweep on the sum of what its
            # kernels need (exact product count x the sampled mix of launches per product)
            'kernels': kernel_rows,
            'sweep_algorithmic_bytes': sweep_bytes,
            'sweep_frac': (sweep_bytes / (elapsed / max(args.steps, 1)) / 1e9 / HBM_PEAK_GBS
                           if elapsed > 0 else None),
            # the per-SNP passes, bracketed the same way
            'snp_pass_avg_ms': {k: prof[k][0] / prof[k][1] for k in
                                ('snp_pass_eval', 'snp_pass_trial', 'snp_pass_trial2',
                                 'snp_pass_trial_lazy', 'snp_pass_trial2_lazy', 'sums_pass',
                                 'sums_pass_store') if prof.get(k, (0, 0))[1]},
        },
    }
    out['collective'] = engine.collective
    out['rccl_ranks'] = rccl_ranks
    if world > 1:
        # per rank: wall time of the timed region, LD bytes of its shard, its dominant kernel
        mine = {'ms_per_step': 1e3 * elapsed_local / args.steps, 'ld_algorithmic_bytes': alg_launch,
                'snps': int(shard.N), 'avg_launch_ms': avg_ms, 'launches': int(launches),
                'achieved_GBps': achieved}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        out['per_rank'] = {k: [r[k] for r in allr] for k in mine}
        ms = out['per_rank']['ms_per_step']
        out['ms_per_step_min_rank'], out['ms_per_step_max_rank'] = min(ms), max(ms)
        slow = int(np.argmax(out['per_rank']['avg_launch_ms']))
        r = allr[slow]
        out['roofline_slowest_rank'] = {
            'rank': slow, 'kernel': dom, 'achieved': r['achieved_GBps'], 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': r['achieved_GBps'] / HBM_PEAK_GBS,
            'avg_launch_ms': r['avg_launch_ms'], 'launches': r['launches'],
            'algorithmic_bytes_per_launch': r['ld_algorithmic_bytes']}
    if world == 1 and not args.no_cpu_baseline and not rehearsal:
        try:
            frac = args.cpu_frac if args.cpu_frac is not None else (0.02 if args.workload == 'C5' else 0.10)
            out['cpu_baseline'] = cpu_baseline(args.workload, args.seed, frac, args.cpu_sweeps,
                                               args.cpu_budget, args.learn_scaling)
        except Exception as exc:      # the GPU number stands on its own
            out['cpu_baseline'] = {'value': None, 'unit': 'sweeps/s', 'cores': 0,
                                   'kind': 'port', 'sample': 'failed: %r' % (exc,)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    engine.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
