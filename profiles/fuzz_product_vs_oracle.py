#!/usr/bin/env python3
"""Seeded random problems across the kernel template space (the generator of
tests/test_gpu_edge_cases.py::test_random_configurations, other seeds): the product on the GPU
against the oracle (oracle/vi.py) sweep by sweep -- ELBO 1e-9 relative, L equal, posterior means,
hyper_delta, error_scaling (tests/test_gpu_edge_cases.py::_compare).

    python profiles/fuzz_product_vs_oracle.py [--seeds 100] [--first 0] [--sweeps 3]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=100)
    ap.add_argument('--first', type=int, default=0)
    ap.add_argument('--sweeps', type=int, default=3)
    ap.add_argument('--shapes', action='store_true',
                    help='round 5: each problem under a work-item shape of its own -- VILMA_LD_TILE (strip '
                         'shapes of the tiled dense product, 0 = round 4\'s kernel) and VILMA_EIG_SLAB_ELEMS '
                         '(many small slabs ... one slab per block = the direct path) drawn from the seed -- '
                         'and blocks of up to 1 300 SNPs in the pool, so that tiles of several slabs, '
                         'off-diagonal tiles and multi-slab eigen blocks occur')
    args = ap.parse_args()
    from test_gpu_edge_cases import _problem, _compare
    pool = [1, 2, 5, 31, 64, 127, 128, 129, 200, 255, 256, 257, 300, 513]
    if args.shapes:
        pool += [511, 512, 640, 700, 1025, 1300]
    bad, t0 = 0, time.time()
    for seed in range(args.first, args.first + args.seeds):
        rng = np.random.default_rng(5000 + seed)
        P = int(rng.integers(1, 6))
        M = int(rng.choice([2, 3, 7, 12, 33, 70, 130]))
        A = int(rng.integers(1, 4))
        sizes = [[int(v) for v in rng.choice(pool, size=int(rng.integers(1, 5)))] for _ in range(P)]
        N = max(sum(s) for s in sizes) + int(rng.integers(0, 9))
        kw = dict(scaled=bool(rng.integers(0, 2)), scale_se=bool(rng.integers(0, 2)))
        label = 'seed %d: P=%d M=%d A=%d N=%d %s' % (seed, P, M, A, N, kw)
        if args.shapes:         # (read by vilma_create: every problem makes its own context)
            srng = np.random.default_rng(9000 + seed)
            tile = str(srng.choice(['auto', '512,4', '512,2', '512,1', '256,2', '256,1', '128,1', '0']))
            slab = int(srng.choice([1024, 4096, 32768, 393216]))
            os.environ.pop('VILMA_LD_TILE', None)
            if tile != 'auto':
                os.environ['VILMA_LD_TILE'] = tile
            os.environ['VILMA_EIG_SLAB_ELEMS'] = str(slab)
            label += ' tile %s slab %d' % (tile, slab)
        try:
            pr = _problem(rng, P, sizes, N=N, M=M, A=A, ldthresh=float(rng.choice([1.0, 0.7])),
                          empty_annot=bool(rng.integers(0, 2)))
            _compare(pr, sweeps=args.sweeps, **kw)
            if seed % 10 == 0:
                print('ok', label, '(%.0f s)' % (time.time() - t0), flush=True)
        except Exception as exc:
            bad += 1
            print('FAIL', label, repr(exc)[:400], flush=True)
    print('%d problems, %d failures, %.0f s' % (args.seeds, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
