#!/usr/bin/env python3
"""End-to-end `vilma fit` through the CLI on a generated on-disk LD schema (2 cohorts, different
block partitions per cohort, 2% flipped alleles, 3% SNPs without LD, 2% without sumstats):
times loading (lazy + threaded eigh), upload and the fit.   python profiles/cli_scale_check.py
[--blocks 150] [--out /tmp/vilma_cli_check]"""
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--blocks', type=int, default=150)
    ap.add_argument('--out', default='/tmp/vilma_cli_check')
    ap.add_argument('--num-its', type=int, default=30)
    ap.add_argument('-K', type=int, default=3)
    ap.add_argument('--only-write', action='store_true',
                    help='generate the on-disk schema and stop (no GPU needed)')
    args = ap.parse_args()
    from vilma_amd.synthetic import block_sizes, ar1_numpy
    rng = np.random.default_rng(0)
    os.makedirs(args.out, exist_ok=True)
    sizes = block_sizes(520 * args.blocks, args.blocks, None, 1)
    n_ld = int(sizes.sum())
    n_extra = int(0.03 * n_ld)
    N = n_ld + n_extra
    ids = np.array(['rs%d' % i for i in range(N)])
    a1 = rng.choice(list('ACGT'), size=N)
    a2 = np.array([rng.choice([x for x in 'ACGT' if x != a]) for a in a1])
    pd.DataFrame({'ID': ids, 'A1': a1, 'A2': a2}).to_csv(os.path.join(args.out, 'extract.tsv'),
                                                         sep='\t', index=False)
    ld_ids = rng.permutation(N)[:n_ld]
    ld_ids.sort()
    t0 = time.perf_counter()
    for p in range(2):
        # cohort 1 uses a shifted partition: blocks of cohort 0 merged pairwise
        psizes = sizes if p == 0 else np.add.reduceat(sizes, np.arange(0, len(sizes), 2))
        lines, lo = [], 0
        for b, n in enumerate(psizes):
            idx = ld_ids[lo:lo + n]
            lo += n
            flip = rng.random(n) < 0.02
            va1 = np.where(flip, a2[idx], a1[idx])
            va2 = np.where(flip, a1[idx], a2[idx])
            var = pd.DataFrame({'ID': ids[idx], 'CHROM': 1, 'BP': idx, 'CM': 0.0, 'A1': va1,
                                'A2': va2})
            var.to_csv(os.path.join(args.out, 'c%d_b%d.var' % (p, b)), sep='\t', header=False,
                       index=False)
            np.save(os.path.join(args.out, 'c%d_b%d.npy' % (p, b)),
                    ar1_numpy(int(n), rng.uniform(0.5, 0.95)))
            lines.append('c%d_b%d.var\tc%d_b%d.npy' % (p, b, p, b))
        open(os.path.join(args.out, 'c%d.schema' % p), 'w').write('\n'.join(lines) + '\n')
        se = rng.uniform(0.005, 0.02, size=N)
        beta = np.where(rng.random(N) < 0.05, rng.normal(0, 0.01, size=N), 0.0)
        bhat = beta + se * rng.normal(size=N)
        keep = rng.random(N) > 0.02
        pd.DataFrame({'ID': ids[keep], 'A1': a1[keep], 'A2': a2[keep], 'BETA': bhat[keep],
                      'SE': se[keep]}).to_csv(os.path.join(args.out, 'sumstats%d.tsv' % p),
                                              sep='\t', index=False)
    print('wrote schema: N=%d SNPs, %d + %d blocks, %.1f s' % (N, len(sizes), len(psizes),
                                                               time.perf_counter() - t0))
    if args.only_write:
        return
    from vilma_amd import frontend
    import logging
    argv = ['fit', '--ld-schema', '%s/c0.schema,%s/c1.schema' % (args.out, args.out),
            '--sumstats', '%s/sumstats0.tsv,%s/sumstats1.tsv' % (args.out, args.out),
            '--extract', '%s/extract.tsv' % args.out, '--output', '%s/run' % args.out,
            '-K', str(args.K), '--num-its', str(args.num_its), '--names', 'eur,eas',
            '--ldthresh', '0.9', '--learn-scaling']
    t0 = time.perf_counter()
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    frontend.main(argv)
    pr.disable()
    print('vilma fit: %.1f s total' % (time.perf_counter() - t0))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(
        'vi_options|load.py|variational_inference|matrix_structures|engine|ld_device|npyio|frame')
    for line in s.getvalue().splitlines():
        if any(k in line for k in ('load_ld_from_schema', 'load_sumstats', '__init__', 'optimize',
                                   'materialize', 'device_blocks', 'load_ld', '_initialize',
                                   'ridge_inverse_dot', 'main', 'stream_cohort', 'ridge_start',
                                   '_load_on_device', 'savez', 'vi_sigma', 'to_csv')):
            print(line[:150])
    out = np.load('%s/run.npz' % args.out)
    est = pd.read_csv('%s/run.estimates.tsv' % args.out, sep='\t')
    print({k: out[k].shape for k in out.files})
    print(est.iloc[:3].to_string())
    print('missing LD eur/eas:', int(est.missing_LD_eur.sum()), int(est.missing_LD_eas.sum()),
          ' error_scaling', out['error_scaling'])


if __name__ == '__main__':
    main()
