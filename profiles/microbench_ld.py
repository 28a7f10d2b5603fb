#!/usr/bin/env python3
"""Micro-benchmark of the LD-product kernels alone on C3-shaped blocks (random symmetric data,
no sumstats/VI setup):  python profiles/microbench_ld.py [--form dense|eig] [--iters 20]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--form', default='dense')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rank-frac', type=float, default=0.28)
    ap.add_argument('--workload', default='C3')
    ap.add_argument('--shard', type=int, default=1, help='use blocks [0, B/shard) only')
    args = ap.parse_args()
    import torch
    from vilma_amd.engine import HipEngine
    from vilma_amd.synthetic import WORKLOADS, block_sizes
    cfg = WORKLOADS[args.workload]
    sizes = block_sizes(cfg['n_ld'], cfg['B'], cfg['fixed'], 0)
    sizes = sizes[:len(sizes) // args.shard]
    P, N = cfg['P'], int(sizes.sum())
    dev = torch.device('cuda', 0)
    nmax = int(sizes.max())
    buf = torch.rand(nmax * nmax, dtype=torch.float64, device=dev)
    svec = torch.rand(nmax, dtype=torch.float64, device=dev)
    eng = HipEngine(P, N, 4, 1)
    perm = np.arange(N, dtype=np.int64)
    for p in range(P):
        if args.form == 'dense':
            blocks = (('dense', buf[:n * n].view(n, n)) for n in sizes)
            specs = [('dense', int(n), int(n)) for n in sizes]
        else:
            rk = [max(1, int(round(args.rank_frac * n))) for n in sizes]
            blocks = (('eig', buf[:n * r].view(n, r), svec[:r]) for n, r in zip(sizes, rk))
            specs = [('eig', int(n), int(r)) for n, r in zip(sizes, rk)]
        eng.load_ld(p, blocks, perm, N, specs=specs)
    x = torch.rand(P, N, dtype=torch.float64, device=dev)
    y = torch.zeros_like(x)
    for _ in range(3):
        eng.ld_matvec_device(x, y)
    torch.cuda.synchronize()
    eng.prof_enable(True)
    eng.prof_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        eng.ld_matvec_device(x, y)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.iters
    prof = eng.prof_read()
    alg, stored = eng.ld_bytes()
    try:
        sms, sbytes = eng.stream_store(5)
        print('bare read of the same store: %.3f ms = %.0f GB/s' % (sms, sbytes / sms / 1e6))
    except Exception as exc:
        print('bare read unavailable:', exc)
    print('form=%s  algorithmic %.3f GB  stored %.3f GB  wall/product %.3f ms'
          % (args.form, alg / 1e9, stored / 1e9, wall * 1e3))
    for k, (ms, n) in prof.items():
        if n:
            per = ms / n
            print('  %-24s %5d launches  avg %.4f ms' % (k, n, per))
    tot = sum(ms for ms, n in prof.values()) / args.iters
    print('  per product: %.4f ms -> algorithmic %.0f GB/s, stored-bytes %.0f GB/s'
          % (tot, alg / tot / 1e6, stored / tot / 1e6))


if __name__ == '__main__':
    main()
