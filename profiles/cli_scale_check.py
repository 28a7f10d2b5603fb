#!/usr/bin/env python3
"""End-to-end `vilma fit` through the CLI on a generated ON-DISK LD schema at the headline size:
where a user's wall time goes (file reads, eigendecompositions, upload, ridge start, sweeps, output).

    python profiles/cli_scale_check.py [--n-ld 1000000 --blocks 1700] [-K 12] [--num-its 1000]
                                       [--ldthresh 1.0] [--out /tmp/vilma_cli_check] [--keep]

Two cohorts on one block partition (BASELINE configs[2]: LDetect-sized AR(1) blocks, the law of
vilma_amd.synthetic.block_sizes), 2 % flipped alleles, 3 % of the SNPs without LD, 2 % without
summary statistics; `.var` + square `.npy` per block, one manifest per cohort (reference load.py:237-354).
The fit is the CLI's own (`vilma_amd.frontend.main`), default mixture grid -K 12 unless told otherwise,
under cProfile; the report lists the cumulative time of the functions a maintainer would look at."""
import argparse
import cProfile
import io
import os
import pstats
import shutil
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_problem(args):
    from vilma_amd.synthetic import block_sizes, ar1_numpy
    rng = np.random.default_rng(0)
    os.makedirs(args.out, exist_ok=True)
    sizes = block_sizes(args.n_ld, args.blocks, None, 0)
    n_ld = int(sizes.sum())
    N = n_ld + int(0.03 * n_ld)
    ids = np.char.add('rs', np.arange(N).astype(str))
    codes = np.array(list('ACGT'))
    i1 = rng.integers(0, 4, size=N)
    i2 = (i1 + rng.integers(1, 4, size=N)) % 4
    a1, a2 = codes[i1], codes[i2]
    pd.DataFrame({'ID': ids, 'A1': a1, 'A2': a2}).to_csv(os.path.join(args.out, 'extract.tsv'),
                                                         sep='\t', index=False)
    ld_ids = np.sort(rng.permutation(N)[:n_ld])
    t0 = time.perf_counter()
    npy_bytes = 0
    for p in range(2):
        lines, lo = [], 0
        for b, n in enumerate(sizes):
            idx = ld_ids[lo:lo + n]
            lo += n
            flip = rng.random(n) < 0.02
            var = pd.DataFrame({'ID': ids[idx], 'CHROM': 1, 'BP': idx, 'CM': 0.0,
                                'A1': np.where(flip, a2[idx], a1[idx]),
                                'A2': np.where(flip, a1[idx], a2[idx])})
            var.to_csv(os.path.join(args.out, 'c%d_b%d.var' % (p, b)), sep='\t', header=False, index=False)
            mat = ar1_numpy(int(n), rng.uniform(0.5, 0.95))
            npy_bytes += mat.nbytes
            np.save(os.path.join(args.out, 'c%d_b%d.npy' % (p, b)), mat)
            lines.append('c%d_b%d.var\tc%d_b%d.npy' % (p, b, p, b))
        open(os.path.join(args.out, 'c%d.schema' % p), 'w').write('\n'.join(lines) + '\n')
        se = rng.uniform(0.005, 0.02, size=N)
        beta = np.where(rng.random(N) < 0.05, rng.normal(0, 0.01, size=N), 0.0)
        bhat = beta + se * rng.normal(size=N)
        keep = rng.random(N) > 0.02
        pd.DataFrame({'ID': ids[keep], 'A1': a1[keep], 'A2': a2[keep], 'BETA': bhat[keep],
                      'SE': se[keep]}).to_csv(os.path.join(args.out, 'sumstats%d.tsv' % p),
                                              sep='\t', index=False)
    print('wrote the problem: N = %d SNPs (%d with LD), %d blocks x 2 cohorts, %.2f GB of .npy, %.1f s'
          % (N, n_ld, len(sizes), npy_bytes / 1e9, time.perf_counter() - t0), flush=True)
    return N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n-ld', type=int, default=1_000_000)
    ap.add_argument('--blocks', type=int, default=1700)
    ap.add_argument('--out', default='/tmp/vilma_cli_check')
    ap.add_argument('--num-its', type=int, default=1000)
    ap.add_argument('-K', type=int, default=12)
    ap.add_argument('--ldthresh', type=float, default=1.0)
    ap.add_argument('--learn-scaling', action='store_true')
    ap.add_argument('--reuse', action='store_true', help='the problem is already in --out')
    ap.add_argument('--keep', action='store_true', help='leave the files in --out')
    ap.add_argument('--top', type=int, default=45)
    args = ap.parse_args()
    if not args.reuse:
        write_problem(args)
    os.system('df -h %s | tail -1' % args.out)
    from vilma_amd import frontend
    argv = ['fit', '--ld-schema', '%s/c0.schema,%s/c1.schema' % (args.out, args.out),
            '--sumstats', '%s/sumstats0.tsv,%s/sumstats1.tsv' % (args.out, args.out),
            '--extract', '%s/extract.tsv' % args.out, '--output', '%s/run' % args.out,
            '-K', str(args.K), '--num-its', str(args.num_its), '--names', 'eur,eas',
            '--ldthresh', str(args.ldthresh)] + (['--learn-scaling'] if args.learn_scaling else [])
    print('vilma ' + ' '.join(argv), flush=True)
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    frontend.main(argv)
    pr.disable()
    total = time.perf_counter() - t0
    print('vilma fit: %.1f s total' % total)
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats('cumulative')
    st.print_stats('vilma_amd|npyio|format.py|linalg|to_csv|savez|readers.py|pickle|zipfile', args.top)
    keep = False
    for line in s.getvalue().splitlines():
        if 'ncalls' in line:
            keep = True
        if keep and line.strip():
            print(line[:170])
    out = np.load('%s/run.npz' % args.out)
    print({k: out[k].shape for k in out.files})
    print('output files: %s' % {f: '%.2f GB' % (os.path.getsize(os.path.join(args.out, f)) / 1e9)
                                for f in os.listdir(args.out) if f.startswith('run')})
    est = pd.read_csv('%s/run.estimates.tsv' % args.out, sep='\t', nrows=3)
    print(est.to_string())
    print('error_scaling', out['error_scaling'])
    if not args.keep:
        shutil.rmtree(args.out, ignore_errors=True)


if __name__ == '__main__':
    main()
