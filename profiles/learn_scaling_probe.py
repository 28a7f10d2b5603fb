"""When does --learn-scaling act?  Sweeps of a workload (or a block sample of it) with scale_se=True
until the error scaling tau has moved `--after` times: per sweep the ELBO gain, tau, wall time.

    python profiles/learn_scaling_probe.py --workload C3 [--blocks 700 870] [--max-sweeps 600]

(_update_error_scaling, reference variational_inference.py:441-448, 472-486, runs in a sweep whose
beta + hyper updates gained less than EM_TOL = 10 in all.)"""
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
    ap.add_argument('--workload', default='C3')
    ap.add_argument('--blocks', type=int, nargs=2, default=None)
    ap.add_argument('--max-sweeps', type=int, default=600)
    ap.add_argument('--after', type=int, default=5, help='stop this many sweeps after tau first moved')
    ap.add_argument('--every', type=int, default=10, help='print every k-th sweep until tau moves')
    a = ap.parse_args()
    import torch
    from test_gpu_fullsize import _setup
    sh, eng, drv = _setup(a.workload, block_range=tuple(a.blocks) if a.blocks else None, scale_se=True)
    drv.initialize_from(sh.fake_mu)
    print('# %s blocks %s: N %d P %d M %d; start ELBO %.6f' % (a.workload, a.blocks, sh.N, sh.P, sh.M, drv._objective))
    state, prev, moved_at = None, drv._objective, None
    t_all = time.perf_counter()
    for it in range(a.max_sweeps):
        ev0 = drv.n_evaluations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        state, _ = drv.sweep(state, lookahead=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        tau = np.array(drv.error_scaling)
        moved = bool(np.any(tau != 1.0))
        if moved and moved_at is None:
            moved_at = it
        if moved or it % a.every == 0:
            print('sweep %4d  gain %14.6f  tau %s  points %2d  %.3f ms' % (
                it, state['elbo'] - prev, np.array2string(tau, precision=10), drv.n_evaluations - ev0, ms), flush=True)
        prev = state['elbo']
        if moved_at is not None and it >= moved_at + a.after:
            break
    print('# tau first moved in sweep %s; %d sweeps in %.2f s; stages ahead %d, skipped %d'
          % (moved_at, it + 1, time.perf_counter() - t_all, drv.n_stages_ahead, drv.n_stages_skipped))
    eng.close()


if __name__ == '__main__':
    main()
