"""rocSOLVER's strided-batched symmetric eigensolvers (called directly through ctypes) against one
torch.linalg.eigh per matrix, on AR(1)-like LD blocks.  Decides whether the loader should batch
same-size-class blocks on the GPU (profiles/README.md).
    python profiles/microbench_batched_eigh.py"""
import ctypes as C
import sys
import time

import numpy as np
import torch

dev = torch.device('cuda', 0)
rs = C.CDLL('librocsolver.so.0')
rb = C.CDLL('librocblas.so.5') if False else rs      # rocsolver re-exports the handle API through rocblas
try:
    rb = C.CDLL('librocblas.so')
except OSError:
    import glob
    rb = C.CDLL(sorted(glob.glob('/opt/rocm/lib/librocblas.so*'))[0])
handle = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(handle)) == 0
EVECT_ORIGINAL, FILL_LOWER, ESORT_ASC = 211, 122, 192

rs.rocsolver_dsyevd_strided_batched.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                C.c_void_p, C.c_int]
rs.rocsolver_dsyevj_strided_batched.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]


def blocks(n, B, seed=0):
    rng = np.random.default_rng(seed)
    idx = np.arange(n)
    out = np.stack([rng.uniform(0.5, 0.95) ** np.abs(idx[:, None] - idx[None, :]) for _ in range(B)])
    return torch.as_tensor(out, device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


for n, B in ((128, 64), (256, 64), (512, 32), (1024, 8)):
    A0 = blocks(n, B)
    w_ref, _ = torch.linalg.eigh(A0[0])

    def single():
        for b in range(B):
            torch.linalg.eigh(A0[b])
    t_single = timed(single)

    D = torch.empty(B, n, dtype=torch.float64, device=dev)
    E = torch.empty(B, n, dtype=torch.float64, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    A = A0.clone()

    def syevd():
        A.copy_(A0)
        st = rs.rocsolver_dsyevd_strided_batched(handle, EVECT_ORIGINAL, FILL_LOWER, n, A.data_ptr(), n,
                                                 n * n, D.data_ptr(), n, E.data_ptr(), n, info.data_ptr(), B)
        assert st == 0, st
    t_syevd = timed(syevd)
    err_d = float((D[0] - w_ref).abs().max())
    rec = (A[0].T * D[0]) @ A[0]            # column-major eigenvectors: A[0] holds V^T in row-major terms
    err_rec_d = float((rec - A0[0]).abs().max())

    resid = torch.empty(B, dtype=torch.float64, device=dev)
    nsw = torch.zeros(B, dtype=torch.int32, device=dev)
    W = torch.empty(B, n, dtype=torch.float64, device=dev)

    def syevj():
        A.copy_(A0)
        st = rs.rocsolver_dsyevj_strided_batched(handle, ESORT_ASC, EVECT_ORIGINAL, FILL_LOWER, n, A.data_ptr(),
                                                 n, n * n, 0.0, resid.data_ptr(), 100, nsw.data_ptr(),
                                                 W.data_ptr(), n, info.data_ptr(), B)
        assert st == 0, st
    try:
        t_syevj = timed(syevj, reps=1)
        err_j = float((W[0] - w_ref).abs().max())
    except AssertionError as exc:
        t_syevj, err_j = float('nan'), float('nan')
    print('n=%4d batch=%3d  torch one-by-one %.2f ms/matrix | syevd batched %.2f ms/matrix (eig err %.1e, '
          'recon err %.1e) | syevj batched %.2f ms/matrix (eig err %.1e, sweeps %d)'
          % (n, B, 1e3 * t_single / B, 1e3 * t_syevd / B, err_d, err_rec_d, 1e3 * t_syevj / B, err_j,
             int(nsw.max())), flush=True)
