#!/usr/bin/env python3
"""Where does a wave of snp_pass_kernel spend its life?  With the -DSNP_TRACE=1 build of the library
every wave leaves seven core-clock stamps (s_memtime); this prints, for the evaluation pass and the
two-step trial at a C3-shaped state (P = 2, N = 1.05 M, M = 40 by default), the distribution of the
phases between the stamps:

    entry -> loop start   per-SNP constants, first vi_mu batch requested
    loop                  this wave's components (M / 4), double-buffered batches
    -> normaliser known   Z parts through LDS + barrier (waiting for the slowest of the four waves)
    -> tile sums          responsibility sums of the tile from the stash (butterflies)
    -> hand-over          barrier, waves 1-3 give their sums to wave 0
    -> exit               wave 0: per-SNP results, objective partials, convergence statistics

    python profiles/snp_pass_timeline.py --build     (where hipcc is: builds libvilma_hip_snptrace.so)
    VILMA_HIP_LIB=vilma_amd/libvilma_hip_snptrace.so python profiles/snp_pass_timeline.py [--M 40]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'vilma_amd', 'libvilma_hip_snptrace.so')


def report(rows, label, ms):
    rows = rows[rows[:, 6] > 0]
    wave = np.arange(len(rows)) % 4
    t = rows[:, :7] - rows[:, :1]
    names = ['entry -> loop start', 'loop', '-> normaliser known', '-> tile sums', '-> hand-over', '-> exit']
    print('[%s] %d waves, kernel %.1f us by events' % (label, len(rows), ms * 1e3))
    real = (rows[:, 9] - rows[:, 8]) * 1e-2             # us on the 100 MHz clock
    ghz = (rows[:, 6] - rows[:, 0]) / np.maximum(real, 1e-2) * 1e-3
    print('  core clock the waves saw (s_memtime cycles per s_memrealtime microsecond): p10 %.2f  p50 %.2f  '
          'p90 %.2f GHz' % tuple(np.percentile(ghz[real > 2.0], [10, 50, 90])))
    for sel, sname in ((wave == 0, 'wave 0'), (wave > 0, 'waves 1-3')):
        d = np.diff(t[sel], axis=1)
        life = t[sel][:, 6]
        print('  %s: life p50 %.0f cycles (p10 %.0f, p90 %.0f)' % (sname, np.median(life), *np.percentile(life, [10, 90])))
        for j, nm in enumerate(names):
            print('    %-22s p50 %7.0f  mean %7.0f  p90 %7.0f   (%4.1f %% of the life)'
                  % (nm, np.median(d[:, j]), d[:, j].mean(), np.percentile(d[:, j], 90),
                     100.0 * d[:, j].mean() / life.mean()))
    # workgroups in flight per CU and the time a slot stays empty between two of them, on the
    # chip-wide 100 MHz clock (10 ns steps)
    full = len(rows) // 4 * 4
    wg = rows[:full].reshape(-1, 4, 12)
    start = wg[:, :, 8].min(axis=1) * 1e-2          # us
    end = wg[:, :, 9].max(axis=1) * 1e-2
    t0 = start.min()
    start -= t0
    end -= t0
    hw = wg[:, 0, 10].astype(np.int64)
    cu = (wg[:, 0, 7].astype(np.int64) << 8) | ((hw >> 8) & 0xff)       # XCC, SE, SH, CU
    span = end.max()
    print('  workgroup life (first entry to last exit of its waves) p50 %.2f us; launch span %.1f us'
          % (np.median(end - start), span))
    cus = np.unique(cu)
    inflight = (end - start).sum() / span / len(cus)
    print('  %d CUs seen; workgroups in flight per CU, time-averaged over the span: %.2f' % (len(cus), inflight))
    gaps, conc_mid = [], []
    for k in cus[:64]:
        m = cu == k
        s_k, e_k = np.sort(start[m]), np.sort(end[m])
        # the i-th start fills the slot the (i - cap)-th end freed: gap = start[i] - end[i - cap]
        mid = (s_k > 0.25 * span) & (s_k < 0.75 * span)
        conc = np.array([((start[m] <= t) & (end[m] > t)).sum() for t in s_k[mid]])
        conc_mid.append(conc.mean() if len(conc) else 0.0)
        cap = int(round(np.percentile(conc, 90))) if len(conc) else 1
        for i in range(cap, len(s_k)):
            if 0.25 * span < s_k[i] < 0.75 * span:
                gaps.append(s_k[i] - e_k[i - cap])
    gaps = np.array(gaps)
    print('  mid-launch: workgroups running on a CU when one more starts: mean %.2f; a start follows the '
          'exit that made room by p50 %.2f us (p10 %.2f, p90 %.2f)'
          % (float(np.mean(conc_mid)), np.median(gaps), *np.percentile(gaps, [10, 90])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--P', type=int, default=2)
    ap.add_argument('--N', type=int, default=1052632)
    ap.add_argument('--M', type=int, default=40)
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    if args.build:
        from vilma_amd import build
        build.build_library(extra_flags=['-DSNP_TRACE=1'], out=LIB, verbose=False)
        return
    import torch
    from vilma_amd.engine import HipEngine
    from vilma_amd.synthetic import mixture_covs
    P, N, M = args.P, args.N, args.M
    rng = np.random.default_rng(0)
    eng = HipEngine(P, N, M, 1)
    se = rng.uniform(0.005, 0.02, size=(P, N))
    eng.set_snp_data(rng.normal(size=(P, N)) / se, se, 1.0 / se ** 2, np.ones((P, N)),
                     np.zeros(N, dtype=np.int64))
    covs = mixture_covs(P, M)
    eng.set_mixture(np.linalg.inv(covs), np.linalg.slogdet(covs)[1])
    eng.set_hyper(np.full((1, M), 1.0 / M))
    for p in range(P):
        eng.load_ld(p, [('dense', np.eye(64))], np.arange(N, dtype=np.int64), 64)
    eng.set_mu(rng.normal(size=(M, P, N)) * 1e-4)
    eng.eval(); eng.accept(False)
    eng.eval(); eng.accept(False)        # the reference normaliser is now this state's own
    eng.synchronize()
    buf = torch.zeros(((N + 63) // 64 * 4, 12), dtype=torch.float64, device='cuda')

    def timed(fn, key):
        for _ in range(3):
            fn()
        eng.synchronize()
        eng.prof_enable(True)
        eng.prof_read(reset=True)
        for _ in range(args.iters):
            fn()
        eng.synchronize()
        got = eng.prof_read()
        eng.prof_enable(False)
        ms, n = got[key]
        return ms / max(n, 1)

    for label, fn, key in (('evaluation', lambda: eng.eval(), 'snp_pass_eval'),
                           ('two-step trial', lambda: eng.trial2(1e-3, 5e-4), 'snp_pass_trial2')):
        ms = timed(fn, key)
        try:
            eng.ld_trace(buf)
        except Exception as exc:
            print('%s: %.4f ms per launch; no trace in this build (%s)' % (label, ms, exc))
            continue
        buf.zero_()
        fn()
        eng.synchronize()
        eng.ld_trace(None)
        report(buf.cpu().numpy(), label, ms)


if __name__ == '__main__':
    main()
