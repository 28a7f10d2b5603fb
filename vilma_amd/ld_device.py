"""Load-time work of a fit on the GPU.

The reference does all of this on the host, once per fit, block by block
(/root/reference/src/vilma/variational_inference.py:189-252 calling matrix_structures.py:
148-152 `dot`, 159-196 `inverse_dot`, 349-387 `ridge_inverse_dot`, 426-447 `diag` / `get_rank`).
At 1 M SNPs x 2 cohorts that is minutes of host time against milliseconds of sweeps, so here
the eigendecompositions are shared between the GPU -- rocSOLVER's BATCHED symmetric eigensolver
through torch.linalg.eigh on stacks of blocks padded to a common size: one launch sequence
serves a whole stack, 0.8 ms per 576-SNP block in stacks of 32 against 13.7 ms alone and 23 ms
on a host thread (profiles/microbench_batched_eigh.py) -- and the host cores (LAPACK `eigh`,
one call per core, the routine the reference uses: still the better place for the small
blocks), and everything that consumes the factors runs on the device:

  * `stream_cohort`: the factors (U, s) of each block travel through pinned staging buffers on
    a copy stream (uploads overlap the host's decompositions and the device work of the block
    before); on the device: diag(R) = (U*U) s, R^+ z = U ((U^T z) / s), chi = z^T R^+ z,
    R R^+ z = U U^T z, and the operator itself -- the dense reconstruction U diag(s) U^T (one
    rocBLAS GEMM, packed into the symmetric slab store by the library) or the eigen form.
    The host never forms an n x n reconstruction and never holds more than a window of blocks.
  * `ridge_start`: (R + diag(reg))^-1 b for ALL blocks of all cohorts at once by Jacobi-
    preconditioned conjugate gradients on the LD store that is already resident -- each
    iteration is one `vilma_ld_matvec` (the hand-written LD product, ~1 ms at 1 M SNPs x 2
    cohorts) -- instead of one Woodbury solve with an r x r inverse per block.  reg =
    se^2 / prior is of order N_snps / (2 N_gwas h^2) >> 0, so the system is well conditioned.
"""
import logging
import time

import numpy as np

from . import matrix_structures as ms


def _factor_fields(block):
    """(u, s) of a materialised block whose operator is u diag(s) u^T (D = 0, v = u^T)."""
    if not np.allclose(block.D, 0):
        raise NotImplementedError('device LD blocks must have a zero diagonal part')
    u, s, v = block.u, block.s, block.v
    if v.shape != u.T.shape or not (np.shares_memory(u, v) or np.array_equal(v, u.T)):
        raise NotImplementedError('device LD blocks must be symmetric factorisations '
                                  '(v == u^T), as LowRankMatrix(X, t) produces')
    return u, s


class _Staging:
    """Pinned host buffers + a copy stream: numpy array -> device tensor without blocking the
    compute stream.  `slots` uploads may be in flight."""

    def __init__(self, torch, device, slots=3):
        self.torch, self.device = torch, device
        self.stream = torch.cuda.Stream(device=device)
        self.slots = [{'buf': None, 'event': None} for _ in range(slots)]
        self.turn = 0

    def upload(self, array):
        """Queue `array` (float64, any shape) for upload; returns (device tensor, event)."""
        t = self.torch
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        n = int(array.size)
        if slot['event'] is not None:
            slot['event'].synchronize()             # the buffer's previous upload has finished
        if slot['buf'] is None or slot['buf'].numel() < n:
            slot['buf'] = t.empty(max(n, 1 << 16), dtype=t.float64).pin_memory()
        host = slot['buf'][:n].view(array.shape)
        host.numpy()[...] = array                    # one host copy into pinned memory
        # The device tensor belongs to the copy stream (allocated under it) and is USED on the
        # compute stream: record_stream keeps the allocator from handing its memory to the next
        # upload while compute kernels that read it are still queued.
        compute = t.cuda.current_stream(self.device)
        with t.cuda.stream(self.stream):
            dev = t.empty(array.shape, dtype=t.float64, device=self.device)
            dev.copy_(host, non_blocking=True)
            ev = t.cuda.Event()
            ev.record(self.stream)
        dev.record_stream(compute)
        slot['event'] = ev
        return dev, ev


def store_upper_bound(lib, sizes, form):
    """Elements to reserve for a cohort's LD store before the ranks are known."""
    total = 0
    for n in sizes:
        n = int(n)
        dense = lib.vilma_ld_dense_elems(n)
        if form == 'dense':
            total += dense
            continue
        if form == 'eig':
            total += lib.vilma_ld_lowrank_elems(n, n)
            continue
        # auto: the eigen form is chosen only up to this rank
        total += max(dense, lib.vilma_ld_lowrank_elems(n, ms.max_eigen_rank(n)))
    return total


# Cost model of one block's eigh, seconds.  Host: LAPACK dsyevd on ONE thread (n = 200 / 588 /
# 1200 / 2431: 3 / 23 / 335 / 1319 ms, profiles/r01f_microbench_eigh.txt).  GPU: rocSOLVER's batched
# dsyevd is launch-bound, so a stack of k equal-size matrices takes about as long as one until the
# chip fills: T1(n) / min(k, ksat(n)) per matrix, T1 = 4 / 14 / 25 / 38 / 66 ms at n = 192 / 576 /
# 1024 / 1536 / 2432 and 0.17 / 0.82 / 2.3 / 6.0 / 18 ms per matrix in stacks of 32
# (profiles/r02s_microbench_batched_eigh.txt).
# `workers` LAPACK threads do not deliver `workers` times one thread (memory-bound tridiagonal
# reductions, numpy's wrappers): 400 blocks of the LDetect size law take 2.7-3.2 s on 16 workers
# against ~12 s of summed single-thread time (profiles/r02s_load_timing.txt)
HOST_POOL_EFFICIENCY = 0.27
EIGH_PAD = 64                 # blocks are padded to a multiple of this to share a stack
EIGH_STACK_BYTES = 1.5e9      # device bytes of one stack (its matrices; rocSOLVER's workspace is extra)
EIGH_STACK_MAX = 64


def _host_eigh_seconds(n):
    return 23e-3 * (n / 588.0) ** 2.9


def _gpu_eigh_seconds(n, stack=1):
    t1 = 4e-3 + 2.7e-5 * n
    ksat = 17.0 * (576.0 / max(n, 1)) ** 1.1
    return t1 / max(1.0, min(float(stack), ksat))


def _padded(n):
    return (int(n) + EIGH_PAD - 1) // EIGH_PAD * EIGH_PAD


def _stack_limit(npad):
    return int(max(1, min(EIGH_STACK_MAX, EIGH_STACK_BYTES // (8 * npad * npad))))


def _stack_fill(npad):
    """Matrices of padded size npad one batched call takes in about the time of one."""
    return 17.0 * (576.0 / max(npad, 1)) ** 1.1


def plan_gpu_eigh(sizes, deferred, workers):
    """Which blocks to eigendecompose on the GPU (in stacks, see gpu_stacks) and which on the
    host pool.  The GPU pass runs before the pool's (see stream_cohort), so a block goes to the
    GPU when its share of a full stack costs less than its share of the pool: in practice
    everything above ~150 SNPs.  Returns a set of block indices (empty when VILMA_GPU_EIGH=0)."""
    import os
    if os.environ.get('VILMA_GPU_EIGH', '1') == '0':
        return set()
    if os.environ.get('VILMA_GPU_EIGH') == 'all':
        return {b for b in range(len(sizes)) if deferred[b]}
    pool = max(1.0, HOST_POOL_EFFICIENCY * workers)
    return {b for b, n in enumerate(sizes)
            if deferred[b] and _gpu_eigh_seconds(_padded(n), _stack_fill(_padded(n)))
            < _host_eigh_seconds(n) / pool}


def gpu_stacks(sizes, on_gpu):
    """Group the GPU's blocks into stacks for the batched eigensolver: largest first, each stack
    padded to its largest member and filled with the next-largest blocks up to the number the
    solver takes in the time of one (launch-bound: _stack_fill), the memory cap, and no member
    smaller than half the padded size.  Returns a list of lists of block indices."""
    order = sorted(on_gpu, key=lambda b: (-sizes[b], b))
    stacks, i = [], 0
    while i < len(order):
        npad = _padded(sizes[order[i]])
        k = int(max(1, min(_stack_limit(npad), np.ceil(_stack_fill(npad)))))
        j = i + 1
        while j < len(order) and j - i < k and 2 * sizes[order[j]] >= npad:
            j += 1
        stacks.append(order[i:j])
        i = j
    return stacks


def _select_from_spectrum(torch, w_dev, Q_cols, w_host, t):
    """The reference's choice of eigenpairs (matrix_structures.select_eigenpairs) applied to an
    ascending spectrum w_host (host) / w_dev (device) with eigenvectors in the COLUMNS of Q_cols
    [n, n] (device view).  Returns (U [n,r] device, s [r] device, s on the host)."""
    n = Q_cols.shape[0]
    idx, degenerate = ms.select_eigenpairs(w_host, t)
    f64 = dict(dtype=torch.float64, device=Q_cols.device)
    if degenerate == 'ones':
        return torch.ones((n, 1), **f64), torch.zeros(1, **f64), np.zeros(1)
    lo, hi = int(idx[0]), int(idx[-1]) + 1
    if hi - lo == idx.size:                          # a contiguous range of the ascending spectrum
        Ud = Q_cols[:, lo:hi].contiguous()
        sd = w_dev[lo:hi].contiguous()
    else:
        sel = torch.as_tensor(idx, device=Q_cols.device)
        Ud = Q_cols[:, sel].contiguous()
        sd = w_dev[sel].contiguous()
    if degenerate == 'zero':
        return Ud, torch.zeros(1, **f64), np.zeros(1)
    return Ud, sd, w_host[idx]


def _gpu_factors_stacked(torch, staging, compute, items):
    """eigh of several symmetric blocks in ONE batched call.  items: list of (key, X [n,n] host,
    t); all are padded to the largest one's size rounded up to EIGH_PAD.  A block sits in the top-left corner of
    its npad x npad slot; the rest of the diagonal is a value below the block's Gershgorin bound
    (-(2 max_i sum_j |X_ij| + 1)), so the padded matrix is block diagonal, its ascending spectrum
    is the padding's eigenvalue `pad` times followed by the block's own, and the block's
    eigenvectors are zero on the padding.  Returns {key: (U [n,r] device, s [r] device, s on the
    host)}."""
    npad = _padded(max(X.shape[0] for _, X, _ in items))
    B = len(items)
    dev = staging.device
    if B == 1 and items[0][1].shape[0] == npad:
        key, X, t = items[0]
        Xd, ev = staging.upload(X)
        compute.wait_event(ev)
        w, Q = torch.linalg.eigh(Xd)
        return {key: _select_from_spectrum(torch, w, Q, w.cpu().numpy(), t)}
    stack = torch.zeros((B, npad, npad), dtype=torch.float64, device=dev)
    for i, (_, X, _) in enumerate(items):
        n = X.shape[0]
        Xd, ev = staging.upload(X)
        compute.wait_event(ev)
        stack[i, :n, :n] = Xd
        if n < npad:
            stack[i].diagonal()[n:] = -(2.0 * Xd.abs().sum(dim=1).max() + 1.0)
    w, Q = torch.linalg.eigh(stack)
    del stack
    w_host = w.cpu().numpy()
    out = {}
    for i, (key, X, t) in enumerate(items):
        n = X.shape[0]
        pad = npad - n
        if not np.all(np.isfinite(w_host[i])):
            raise RuntimeError('batched eigh failed on a block of %d SNPs (non-finite spectrum)' % n)
        out[key] = _select_from_spectrum(torch, w[i, pad:], Q[i, :n, pad:], w_host[i, pad:], t)
    return out


def _gpu_factors(torch, staging, compute, X, t):
    """eigh of one symmetric block on the GPU and the reference's selection of eigenpairs.
    Returns (U [n,r] device, s [r] device, s on the host)."""
    return _gpu_factors_stacked(torch, staging, compute, [(0, X, t)])[0]


def _load_symmetric(m):
    """The dense matrix of a deferred block, checked like LowRankMatrix(X, t) checks it."""
    make_matrix, t = m._thunk
    X = np.ascontiguousarray(make_matrix(), dtype=np.float64)
    if not np.allclose(X, X.T):
        raise ValueError('Provided matrix is not symmetric')
    return X, t


import os as _os
# host threads feeding the batched solver, one stream each (VILMA_GPU_EIGH_LANES overrides)
GPU_EIGH_LANES = int(_os.environ.get('VILMA_GPU_EIGH_LANES', '3'))


def _decompose_on_gpu(torch, device, pool, mats, sizes, on_gpu):
    """Eigendecompose the blocks `on_gpu` stack by stack (gpu_stacks: largest first).  rocSOLVER's
    launch sequences are bound by the host thread that issues them (two threads on two streams:
    1.5x on stacks of blocks up to ~600 SNPs, nothing above 1 000), so GPU_EIGH_LANES threads take
    alternate stacks, each on its own stream with its own staging buffers; the host pool reads and
    checks a lane's next stack while the GPU works on its current one.  Returns {block index:
    (U, s, s on the host)}, everything on the device and safe to use on the current stream."""
    from concurrent.futures import ThreadPoolExecutor
    chunks = gpu_stacks(sizes, on_gpu)
    if not chunks:
        return {}
    compute = torch.cuda.current_stream(device)
    lanes = max(1, min(GPU_EIGH_LANES, len(chunks)))

    def read(chunk):
        return [(b, pool.submit(_load_symmetric, mats[b])) for b in chunk]

    def lane(k):
        mine = chunks[k::lanes]
        stream = torch.cuda.Stream(device=device)
        stream.wait_stream(compute)
        staging = _Staging(torch, device)
        out = {}
        with torch.cuda.stream(stream):
            nxt = read(mine[0])
            for c in range(len(mine)):
                cur, nxt = nxt, (read(mine[c + 1]) if c + 1 < len(mine) else None)
                items = []
                for b, fut in cur:
                    X, t = fut.result()
                    items.append((b, X, t))
                out.update(_gpu_factors_stacked(torch, staging, stream, items))
        stream.synchronize()
        return out

    factors = {}
    if lanes == 1:
        factors.update(lane(0))
    else:
        with ThreadPoolExecutor(max_workers=lanes) as ex:
            for part in ex.map(lane, range(lanes)):
                factors.update(part)
    for Ud, sd, _ in factors.values():          # allocated on a lane's stream, used on this one
        Ud.record_stream(compute)
        sd.record_stream(compute)
    return factors


def stream_cohort(engine, cohort, ld, form, z_ld, workers=None, window=None):
    """Decompose (GPU in stacks, then host thread pool) and install (device) the blocks of one
    cohort's local BlockDiagonalMatrix `ld`, in LD order.  z_ld [n_ld]: z = beta-hat / se at the LD positions.

    Returns dict(diag [n_ld], rmle [n_ld] = R R^+ z, chi = z^T R^+ z, rank) -- what
    VIScheme.__init__ derives per cohort (variational_inference.py:189-192, 236-252)."""
    from concurrent.futures import ThreadPoolExecutor
    torch = engine.torch
    dev = engine.device
    mats = ld.matrices
    sizes = [m.shape[0] for m in mats]
    n_ld = int(np.sum(sizes)) if sizes else 0
    perm = ld.perm.astype(np.int64)
    engine.ld_begin(cohort, len(mats), perm, n_ld,
                    store_upper_bound(engine.lib, sizes, form))
    z_dev = torch.as_tensor(np.ascontiguousarray(z_ld, dtype=np.float64), device=dev)
    diag = torch.zeros(n_ld, dtype=torch.float64, device=dev)
    rmle = torch.zeros(n_ld, dtype=torch.float64, device=dev)
    chi = torch.zeros((), dtype=torch.float64, device=dev)
    rank = 0
    staging = _Staging(torch, dev)
    workers = ms._default_workers() if workers is None else workers
    window = 2 * workers if window is None else window

    on_gpu = plan_gpu_eigh(sizes, [m.is_deferred() for m in mats], workers)

    def decompose(b):
        m = mats[b]
        if b in on_gpu:
            return m, b                      # already decomposed on the GPU (first pass below)
        m.materialize()                  # np.linalg.eigh + thresholding; releases the GIL
        return m, None

    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)        # one LAPACK thread per worker
    except ImportError:
        limiter = None
    compute = torch.cuda.current_stream(dev)
    try:
        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            # first pass: the GPU's share, stack by stack (device-resident factors); second pass:
            # every block in LD order, the host pool decomposing its share a window ahead
            t0 = time.perf_counter()
            gpu_factors = _decompose_on_gpu(torch, dev, pool, mats, sizes, on_gpu)
            t_gpu = time.perf_counter() - t0
            pending = []
            it = iter(range(len(mats)))
            start = 0

            def refill():
                while len(pending) < window:
                    try:
                        pending.append(pool.submit(decompose, next(it)))
                    except StopIteration:
                        return
            refill()
            t_wait = t_dev = 0.0
            while pending:
                t0 = time.perf_counter()
                m, raw = pending.pop(0).result()
                t_wait += time.perf_counter() - t0
                t0 = time.perf_counter()
                refill()
                if raw is not None:
                    Ud, sd, s = gpu_factors.pop(raw)
                    n, r = Ud.shape
                    rank += r if r > 1 else (0 if s[0] == 0 else 1)     # LowRankMatrix.get_rank
                else:
                    u, s = _factor_fields(m)
                    n, r = u.shape
                    rank += m.get_rank()
                    Ud, ev_u = staging.upload(u)
                    sd, ev_s = staging.upload(s)
                    compute.wait_event(ev_u)
                    compute.wait_event(ev_s)
                dense = form == 'dense' or (form == 'auto' and ms.dense_is_cheaper(n, r))
                zb = z_dev[start:start + n]
                inv_s = torch.where(sd != 0, 1.0 / sd, torch.zeros_like(sd))
                proj = Ud.T @ zb                               # U^T z
                chi += (proj * proj * inv_s).sum()             # z^T R^+ z
                rmle[start:start + n] = Ud @ (proj * (sd * inv_s))   # R R^+ z
                diag[start:start + n] = (Ud * Ud) @ sd
                if dense:
                    engine.ld_add(cohort, ('dense', (Ud * sd) @ Ud.T))
                else:
                    engine.ld_add(cohort, ('eig', Ud, sd))
                if m.__dict__.get('_from_thunk'):
                    m.forget()           # the factors live on the device now
                start += n
                t_dev += time.perf_counter() - t0
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    engine.ld_end(cohort)
    logging.info('LD cohort %d: %d blocks streamed to the device (%d decomposed on the GPU in '
                 '%.2f s); main thread waited %.2f s for the host pool, spent %.2f s on uploads '
                 'and device work', cohort, len(mats), len(on_gpu), t_gpu, t_wait, t_dev)
    return {'diag': diag.cpu().numpy(), 'rmle': rmle.cpu().numpy(), 'chi': float(chi.item()),
            'rank': float(rank), 'wait_s': t_wait, 'device_s': t_dev, 'gpu_eigh': len(on_gpu),
            'gpu_eigh_s': t_gpu}


class RidgeStalled(RuntimeError):
    """Conjugate gradients did not get the ridge start below the residual the caller can use."""


RIDGE_ACCEPT = 1e-6     # true relative residual good enough for a starting point that is then
                        # jittered by 1e-3 se (variational_inference.py:643-657)


def ridge_start(engine, b, reg, diag, rtol=1e-13, max_iter=20000):
    """x = (R_p + diag(reg_p))^-1 b_p for every cohort p at once (b, reg, diag: [P, N] in SNP
    order; reg > 0) by Jacobi-preconditioned conjugate gradients on the resident LD store --
    the per-block ridge solve of reference matrix_structures.py:349-387, without factorising
    anything.  The result only seeds _initialize, so an iterate whose TRUE residual is below
    RIDGE_ACCEPT is accepted with a warning when the recurrence stalls above `rtol` (a small
    regulariser on strongly correlated blocks makes 1e-13 tight in fp64); beyond that it raises
    RidgeStalled and the caller falls back to the reference's per-block solve on the host."""
    torch = engine.torch
    dev = engine.device
    f64 = dict(dtype=torch.float64, device=dev)
    b = torch.as_tensor(np.ascontiguousarray(b, dtype=np.float64), **f64)
    reg = torch.as_tensor(np.ascontiguousarray(reg, dtype=np.float64), **f64)
    minv = 1.0 / (torch.as_tensor(np.ascontiguousarray(diag, dtype=np.float64), **f64) + reg)
    P = b.shape[0]
    x = torch.zeros_like(b)
    bnorm = torch.linalg.vector_norm(b, dim=1)
    if float(bnorm.max().item()) == 0.0:
        return x.cpu().numpy()
    live = (bnorm > 0).to(torch.float64)             # a cohort with b = 0 stays at x = 0
    safe = torch.where(bnorm > 0, bnorm, torch.ones_like(bnorm))
    r = b.clone()
    zv = minv * r
    p = zv.clone()
    rz = (r * zv).sum(dim=1)
    Ap = torch.empty_like(b)
    it, worst = 0, float('inf')
    while it < max_iter:
        for _ in range(8):                           # residual check (a host sync) every 8 steps
            engine.ld_matvec_device(p, Ap)
            Ap += reg * p
            pAp = (p * Ap).sum(dim=1)
            alpha = live * rz / torch.where(pAp != 0, pAp, torch.ones_like(pAp))
            x += alpha[:, None] * p
            r -= alpha[:, None] * Ap
            zv = minv * r
            rz_new = (r * zv).sum(dim=1)
            beta = live * rz_new / torch.where(rz != 0, rz, torch.ones_like(rz))
            p = zv + beta[:, None] * p
            rz = rz_new
            it += 1
        worst = float((torch.linalg.vector_norm(r, dim=1) / safe).max().item())
        if worst <= rtol:
            break
    stalled = not worst <= rtol
    # the recurrence residual can drift from the true one: verify against the operator itself
    engine.ld_matvec_device(x, Ap)
    Ap += reg * x
    true_res = float((torch.linalg.vector_norm(b - Ap, dim=1) / safe).max().item())
    if stalled or not true_res <= 1e-9:
        if not true_res <= RIDGE_ACCEPT:
            raise RidgeStalled('ridge start: relative residual %.3e (recurrence %.3e) after %d '
                               'iterations of conjugate gradients' % (true_res, worst, it))
        logging.warning('ridge start: conjugate gradients stopped at relative residual %.3e after '
                        '%d iterations (wanted %.0e); accepted -- it only seeds the jittered '
                        'starting point', true_res, it, rtol)
    return x.cpu().numpy()
