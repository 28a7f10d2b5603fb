#!/usr/bin/env python3
"""Why does ld_sym_kernel stream at 0.85 - 0.97 of a bare read of the same store, depending on the
process (profiles/r03r_placement_probe.txt)?  One process, one placement of a C3-shaped LD store:

  1. bare reads of the store under different access patterns (vilma_prof_stream_pattern): chunk
     size, store order against scattered order, number of workgroups;
  2. ld_sym_kernel with its work items longest-first (the default), in store order, and in store
     order dealt out per XCD (vilma_prof_ld_order), interleaved and repeated;
  3. with the -DLD_TRACE=1 build of the library (VILMA_HIP_LIB=vilma_amd/libvilma_hip_trace.so,
     built by `python profiles/ld_levels_probe.py --build-variants` where hipcc is): the
     per-workgroup trace of one launch per order -- per-XCD bytes, span and finish time, the tail
     of the launch, the spread of per-workgroup streaming rates, the core clock the workgroups saw;
  4. the same bare reads with the thin stream of small stores ld_sym_kernel makes beside them,
     and (other builds made by --build-variants, VILMA_HIP_LIB=.../libvilma_hip_<variant>.so)
     ld_sym_kernel without its stores, with the plain stores of rounds 1 - 3, non-temporal, staged
     but plain, write-through but not staged (the product stages AND writes through).

    python profiles/ld_levels_probe.py [--iters 20] [--workload C3]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, 'vilma_amd', 'libvilma_hip_trace.so')


def build_engine(workload, shard):
    import torch
    from vilma_amd.engine import HipEngine
    from vilma_amd.synthetic import WORKLOADS, block_sizes
    cfg = WORKLOADS[workload]
    sizes = block_sizes(cfg['n_ld'], cfg['B'], cfg['fixed'], 0)
    sizes = sizes[:len(sizes) // shard]
    P, N = cfg['P'], int(sizes.sum())
    dev = torch.device('cuda', 0)
    nmax = int(sizes.max())
    buf = torch.rand(nmax * nmax, dtype=torch.float64, device=dev)
    eng = HipEngine(P, N, 4, 1)
    perm = np.arange(N, dtype=np.int64)
    for p in range(P):
        blocks = (('dense', buf[:n * n].view(n, n)) for n in sizes)
        eng.load_ld(p, blocks, perm, N, specs=[('dense', int(n), int(n)) for n in sizes])
    x = torch.rand(P, N, dtype=torch.float64, device=dev)
    y = torch.zeros_like(x)
    return eng, x, y


def time_ld(eng, x, y, iters):
    import torch
    for _ in range(3):
        eng.ld_matvec_device(x, y)
    torch.cuda.synchronize()
    eng.prof_enable(True)
    eng.prof_read(reset=True)
    for _ in range(iters):
        eng.ld_matvec_device(x, y)
    torch.cuda.synchronize()
    ms, n = eng.prof_read()['ld_sym_kernel']
    eng.prof_enable(False)
    return ms / max(n, 1)


def time_ld_two_rhs(eng, P, N, M, iters):
    """ld_sym_kernel with two right-hand sides (the product of a two-step trial), by the library's
    events; a small mixture keeps the per-SNP pass beside it short."""
    import torch
    from vilma_amd.synthetic import mixture_covs
    rng = np.random.default_rng(1)
    se = rng.uniform(0.005, 0.02, size=(P, N))
    eng.set_snp_data(rng.normal(size=(P, N)) / se, se, 1.0 / se ** 2, np.ones((P, N)),
                     np.zeros(N, dtype=np.int64))
    covs = mixture_covs(P, M)
    eng.set_mixture(np.linalg.inv(covs), np.linalg.slogdet(covs)[1])
    eng.set_hyper(np.full((1, M), 1.0 / M))
    eng.set_mu(rng.normal(size=(M, P, N)) * 1e-4)
    eng.eval(); eng.accept(False)
    for _ in range(3):
        eng.trial2(1e-3, 5e-4)
    torch.cuda.synchronize()
    eng.prof_enable(True)
    eng.prof_read(reset=True)
    for _ in range(iters):
        eng.trial2(1e-3, 5e-4)
    torch.cuda.synchronize()
    ms, n = eng.prof_read()['ld_sym_kernel_two_rhs']
    eng.prof_enable(False)
    return ms / max(n, 1)


def analyse_trace(rows, label):
    """rows [n, 5]: start, end (100 MHz ticks), XCC id, bytes, core-clock cycles start to end."""
    rows = rows[rows[:, 1] > 0]
    ticks = np.maximum(rows[:, 1] - rows[:, 0], 1.0)
    mhz = rows[:, 4] / (ticks * 1e-2)           # cycles per microsecond
    long = ticks >= 2000                        # workgroups that lived 20 us or more
    if long.any():
        q = np.percentile(mhz[long], [1, 50, 99])
        print('  core clock seen by workgroups of 20 us or more (s_memtime / s_memrealtime): '
              'p1 %.0f  p50 %.0f  p99 %.0f MHz' % tuple(q))
    t0 = rows[:, 0].min()
    start = (rows[:, 0] - t0) * 1e-2          # microseconds
    end = (rows[:, 1] - t0) * 1e-2
    xcc = rows[:, 2].astype(int)
    nbytes = rows[:, 3]
    span = end.max()
    print('  [%s] %d workgroups, %.3f GB, launch span %.1f us = %.0f GB/s'
          % (label, len(rows), nbytes.sum() / 1e9, span, nbytes.sum() / span / 1e3))
    order = np.argsort(end)
    cum = np.cumsum(nbytes[order]) / nbytes.sum()
    marks = [0.5, 0.9, 0.99, 1.0]
    print('      bytes done by: ' + ', '.join('%d%% at %.1f us' % (100 * m, end[order][np.searchsorted(cum, m - 1e-12)])
                                            for m in marks))
    print('      last workgroup to START: %.1f us; workgroups running at the end: ' % start.max(), end='')
    for back in (5.0, 20.0, 50.0):
        print('%d within the last %.0f us, ' % (int(((end > span - back)).sum()), back), end='')
    print()
    dur = np.maximum(end - start, 1e-2)
    rate = nbytes / dur / 1e3                  # GB/s per workgroup
    big = nbytes >= 0.5 * nbytes.max()
    q = np.percentile(rate[big], [5, 25, 50, 75, 95])
    print('      per-workgroup rate of the large chunks (GB/s): p5 %.2f  p25 %.2f  p50 %.2f  p75 %.2f  p95 %.2f'
          % tuple(q))
    # how evenly is the chip loaded over the launch: bytes in flight per 50-us slice
    edges = np.arange(0.0, span + 50.0, 50.0)
    mid = 0.5 * (start + end)
    hist, _ = np.histogram(mid, bins=edges, weights=nbytes)
    print('      GB/s by 50-us slice (chunk midpoints): ' + ' '.join('%.0f' % (h / 50.0 / 1e3) for h in hist))
    print('      XCC  workgroups   GB    first start  last end   rate while busy (GB/s)')
    for k in sorted(set(xcc.tolist())):
        m = xcc == k
        busy = end[m].max() - start[m].min()
        print('      %3d  %9d  %6.3f  %9.1f  %9.1f   %8.0f'
              % (k, int(m.sum()), nbytes[m].sum() / 1e9, start[m].min(), end[m].max(),
                 nbytes[m].sum() / busy / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--workload', default='C3')
    ap.add_argument('--shard', type=int, default=1)
    ap.add_argument('--two-rhs', action='store_true',
                    help='only time the product with one and with two right-hand sides (any build)')
    ap.add_argument('--build-variants', action='store_true',
                    help='only build the diagnostic variants of the library next to it and exit: '
                         'libvilma_hip_trace.so (-DLD_TRACE=1) and the LD_STORE_MODE builds of '
                         'kernels.hip: _nostore (1: wrong results, timing only), _plainstore (2: '
                         'rounds 1 - 3), _ntstore (3), _stagedplain (4), _sc1store (5), _staged8 (6: staged, '
                         '8-byte write-through stores)')
    args = ap.parse_args()
    if args.build_variants:
        from concurrent.futures import ThreadPoolExecutor
        from vilma_amd import build
        jobs = [(['-DLD_TRACE=1'], TRACE_LIB),
                (['-DLD_STORE_MODE=1'], TRACE_LIB.replace('_trace', '_nostore')),
                (['-DLD_STORE_MODE=2'], TRACE_LIB.replace('_trace', '_plainstore')),
                (['-DLD_STORE_MODE=3'], TRACE_LIB.replace('_trace', '_ntstore')),
                (['-DLD_STORE_MODE=4'], TRACE_LIB.replace('_trace', '_stagedplain')),
                (['-DLD_STORE_MODE=5'], TRACE_LIB.replace('_trace', '_sc1store')),
                (['-DLD_STORE_MODE=6'], TRACE_LIB.replace('_trace', '_staged8'))]
        with ThreadPoolExecutor(max_workers=7) as pool:
            list(pool.map(lambda j: build.build_library(extra_flags=j[0], out=j[1], verbose=False), jobs))
        return
    import torch
    eng, x, y = build_engine(args.workload, args.shard)
    if args.two_rhs:
        one = time_ld(eng, x, y, args.iters)
        two = time_ld_two_rhs(eng, x.shape[0], x.shape[1], 4, args.iters)
        sms, sbytes = eng.stream_store(5)
        print('library %s: ld_sym_kernel one right-hand side %.4f ms, two %.4f ms (ratio %.3f); bare read of '
              'the store %.4f ms' % (os.environ.get('VILMA_HIP_LIB', 'default'), one, two, two / one, sms))
        return
    alg, stored = eng.ld_bytes()
    print('store %.3f GB (algorithmic %.3f GB symmetric-half basis %.3f GB), library %s'
          % (stored / 1e9, alg / 1e9, alg / 2e9, os.environ.get('VILMA_HIP_LIB', 'default')))
    sms, sbytes = eng.stream_store(5)
    print('bare read (32 KB steps in store order, 4096 workgroups): %.3f ms = %.0f GB/s'
          % (sms, sbytes / sms / 1e6))
    print('access patterns of a bare read (chunk per workgroup, order, grid):')
    for chunk_kb, scattered, grid in ((32, 0, 4096), (32, 1, 4096), (128, 0, 2048), (128, 1, 2048),
                                      (512, 0, 2048), (512, 1, 2048), (512, 1, 1024),
                                      (512, 1, 4096), (2048, 1, 2048)):
        ms, nb = eng.stream_pattern(chunk_kb, scattered, grid)
        print('  chunk %5d KB  %-9s  grid %5d: %.3f ms = %.0f GB/s'
              % (chunk_kb, 'scattered' if scattered else 'in order', grid, ms, nb / ms / 1e6))
    print('the same bare reads with 8 doubles stored per wave per 8 KiB read (chunk, order, grid, stores):')
    for chunk_kb, scattered, grid in ((32, 0, 4096), (512, 1, 2048), (512, 1, 1536)):
        for writes, wname in ((0, 'none'), (1, '64 B / wave / 8 KiB'), (2, 'same, non-temporal'),
                              (3, 'whole lines'), (4, 'half the bytes'),
                              (5, '4 KiB / WG / 512 KB'), (6, 'all at the WG end'),
                              (7, '4 KiB / WG, time order'), (8, '64 B / wave, time order'),
                              (9, '64 B / wave, sc1'), (10, '4 KiB / WG, sc1'),
                              (0, 'none')):
            ms, nb = eng.stream_pattern(chunk_kb, scattered, grid, writes=writes)
            print('  chunk %5d KB  %-9s  grid %5d  stores %-24s: %.3f ms = %.0f GB/s'
                  % (chunk_kb, 'scattered' if scattered else 'in order', grid, wname, ms, nb / ms / 1e6))
    print('ld_sym_kernel by order of its work items (avg ms per launch, %d launches each):' % args.iters)
    names = {0: 'longest first', 1: 'store order', 2: 'store order per XCD'}
    for rep in range(2):
        for order in (0, 1, 2):
            eng.ld_order(order)
            ms = time_ld(eng, x, y, args.iters)
            print('  rep %d  %-20s %.4f ms' % (rep, names[order], ms))
    try:
        buf = torch.zeros((400000, 5), dtype=torch.float64, device=x.device)
        eng.ld_trace(buf)
    except Exception as exc:
        print('no trace in this build (%s)' % exc)
        return
    for order in (0, 1, 2):
        eng.ld_order(order)
        for _ in range(2):
            eng.ld_matvec_device(x, y)
        torch.cuda.synchronize()
        buf.zero_()
        eng.ld_matvec_device(x, y)
        torch.cuda.synchronize()
        analyse_trace(buf.cpu().numpy(), names[order])
    eng.ld_trace(None)


if __name__ == '__main__':
    main()
