"""Load-time block decomposition: numpy.linalg.eigh on the host cores against
torch.linalg.eigh (rocSOLVER) on the GPU, for LDetect-sized AR(1) blocks.

    python3 profiles/microbench_eigh.py [--sizes 200,588,1200,2431] [--threads 16]

Prints one JSON line per size: host seconds per block with 1 thread and with `threads` blocks in
flight, GPU seconds per block (single matrix and a batch of equal-sized ones), and the largest
deviation of eigenvalues / of the reconstruction U diag(s) U^T between the two."""
import argparse
import json
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def ar1(n, rho):
    i = np.arange(n)
    return rho ** np.abs(i[:, None] - i[None, :])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='200,588,1200,2431')
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--batch', type=int, default=8)
    args = ap.parse_args()
    import torch
    dev = torch.device('cuda', 0)
    for n in [int(v) for v in args.sizes.split(',')]:
        mats = [ar1(n, 0.5 + 0.45 * (k + 1) / (args.threads + 1)) for k in range(args.threads)]
        t0 = time.perf_counter()
        s_ref, u_ref = np.linalg.eigh(mats[0])
        host1 = time.perf_counter() - t0
        with ThreadPoolExecutor(args.threads) as pool:
            t0 = time.perf_counter()
            list(pool.map(np.linalg.eigh, mats))
            host_par = (time.perf_counter() - t0) / len(mats)
        a = torch.as_tensor(mats[0], device=dev)
        torch.linalg.eigh(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s_gpu, u_gpu = torch.linalg.eigh(a)
        torch.cuda.synchronize()
        gpu1 = time.perf_counter() - t0
        stack = torch.as_tensor(np.stack(mats[:args.batch]), device=dev)
        torch.linalg.eigh(stack)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.linalg.eigh(stack)
        torch.cuda.synchronize()
        gpub = (time.perf_counter() - t0) / args.batch
        s_g, u_g = s_gpu.cpu().numpy(), u_gpu.cpu().numpy()
        rec_ref = (u_ref * s_ref) @ u_ref.T
        rec_gpu = (u_g * s_g) @ u_g.T
        print(json.dumps({
            'n': n, 'host_1thread_s': host1, 'host_%d_in_flight_s_per_block' % args.threads: host_par,
            'gpu_single_s': gpu1, 'gpu_batch%d_s_per_block' % args.batch: gpub,
            'eigval_max_abs_dev': float(np.abs(s_g - s_ref).max()),
            'recon_max_abs_dev': float(np.abs(rec_gpu - rec_ref).max()),
            'recon_vs_input_gpu': float(np.abs(rec_gpu - mats[0]).max()),
            'recon_vs_input_host': float(np.abs(rec_ref - mats[0]).max()),
        }))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
