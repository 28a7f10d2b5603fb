#!/usr/bin/env python3
"""Micro-benchmark of the per-SNP kernels alone (C3 shape: P=2, N=1.05M, M=40) with a negligible
LD store:  python profiles/microbench_snp.py [--P 2 --N 1052632 --M 40 --iters 20]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--P', type=int, default=2)
    ap.add_argument('--N', type=int, default=1052632)
    ap.add_argument('--M', type=int, default=40)
    ap.add_argument('--A', type=int, default=1)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--step', type=float, default=1e-3,
                    help='trial step: small keeps the candidate near the reference state (no tile is redone)')
    args = ap.parse_args()
    import torch
    from vilma_amd.engine import HipEngine
    from vilma_amd.synthetic import mixture_covs
    P, N, M, A = args.P, args.N, args.M, args.A
    rng = np.random.default_rng(0)
    eng = HipEngine(P, N, M, A)
    se = rng.uniform(0.005, 0.02, size=(P, N))
    eng.set_snp_data(rng.normal(size=(P, N)) / se, se, 1.0 / se ** 2, np.ones((P, N)),
                     rng.integers(0, A, size=N))
    covs = mixture_covs(P, M)
    eng.set_mixture(np.linalg.inv(covs), np.linalg.slogdet(covs)[1])
    eng.set_hyper(np.full((A, M), 1.0 / M))
    for p in range(P):
        eng.load_ld(p, [('dense', np.eye(64))], np.arange(N, dtype=np.int64), 64)
    eng.set_mu(rng.normal(size=(M, P, N)) * 1e-3)
    eng.eval(); eng.accept(False)
    eng.synchronize()

    def timed(fn):
        for _ in range(3):
            fn()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        eng.synchronize()
        return (time.perf_counter() - t0) / args.iters * 1e3
    t_eval = timed(lambda: eng.eval())
    t_trial = timed(lambda: eng.trial(args.step))
    t_trial2 = timed(lambda: eng.trial2(args.step, 0.5 * args.step))
    have = eng.trial_sums_available()
    t_tile = timed(lambda: eng.trial_sums(both=have == 2)) if have else float('nan')
    t_sums = timed(lambda: eng.delta_sums())
    gb = 8e-9 * N * M * P
    print('P=%d N=%d M=%d A=%d' % (P, N, M, A))
    print('  eval  (read mu)        %.3f ms  -> %.0f GB/s' % (t_eval, gb / t_eval * 1e3))
    print('  trial (read+write mu)  %.3f ms  -> %.0f GB/s' % (t_trial, 2 * gb / t_trial * 1e3))
    print('  trial2 (read, write 2) %.3f ms  -> %.0f GB/s' % (t_trial2, 3 * gb / t_trial2 * 1e3))
    print('  trial_sums (%d cand.)   %.3f ms' % (have, t_tile))
    print('  delta_sums             %.3f ms  -> %.0f GB/s' % (t_sums, gb / t_sums * 1e3))
    print('  (each includes the tiny LD product + finalize, ~0.03 ms)')


if __name__ == '__main__':
    main()
