"""Block-diagonal LD containers with the reference's interface (vilma.matrix_structures).

`LowRankMatrix` and `BlockDiagonalMatrix` keep the constructor arguments, fields (`u`, `s`,
`v`, `D`, `inv_s`, `matrices`, `starts`, `perm`, `inv_perm`, `missing`, `shape`) and method
names of /root/reference/src/vilma/matrix_structures.py:38-447 so loaders, tests and callers
written against the reference keep working.  Division of labour on MI355X:

  * `BlockDiagonalMatrix.dot` -- the per-sweep operator -- runs on the GPU through
    libvilma_hip.so (vilma_ld_matvec); there is no host implementation of it.
  * the load-time pieces that the reference runs once per fit (eigendecomposition and
    thresholding, pseudo-inverse, ridge solve, diag, rank: matrix_structures.py:15-28, 159-234,
    349-387, 426-447) are host numpy here; they produce the constants the device needs.
"""
import numpy as np


def _svd_threshold(matrix, ld_thresh):
    """Eigendecompose a symmetric block and keep eigenvalues >= 1 - sqrt(ld_thresh)
    (reference matrix_structures.py:15-28).  Returns (u, s, v); an empty selection gives the
    rank-one zero operator the reference uses as a stand-in."""
    w, q = np.linalg.eigh(matrix)
    sel = np.flatnonzero(w >= 1 - np.sqrt(ld_thresh))
    if sel.size == 0:
        n = matrix.shape[0]
        return np.ones((n, 1)), np.zeros(1), np.ones((1, matrix.shape[1]))
    u = np.array(q[:, sel])
    return u, np.array(w[sel]), np.array(u.T)


def select_eigenpairs(w, ld_thresh):
    """Which eigenpairs of a block the reference keeps, given ALL eigenvalues `w` (any order):
    those >= 1 - sqrt(ld_thresh) (matrix_structures.py:15-28), then of these the ones
    > 1e-12 x the largest kept (:136-146).  Returns (indices into w, degenerate) where
    degenerate is None, or 'ones' (nothing passed the first rule: the rank-one zero operator
    u = 1, s = 0) or 'zero' (nothing passed the second: first kept vector with s = 0)."""
    sel = np.flatnonzero(w >= 1 - np.sqrt(ld_thresh))
    if sel.size == 0:
        return sel, 'ones'
    s = w[sel]
    big = s > 1e-12 * np.max(s)
    if big.any():
        return sel[big], None
    return sel[:1], 'zero'


def _default_workers():
    try:
        import os
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 4


def parallel_map(fn, items, workers=None):
    """map over LD blocks on a thread pool (numpy/LAPACK release the GIL); BLAS is held to one
    thread per worker meanwhile.  Order of results = order of items."""
    items = list(items)
    workers = _default_workers() if workers is None else workers
    if workers <= 1 or len(items) < 2:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except ImportError:
        limiter = None
    try:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            return list(pool.map(fn, items))
    finally:
        if limiter is not None:
            limiter.restore_original_limits()


def dense_bytes_moved(n):
    """Elements the dense symmetric kernel reads per product: the lower triangle by 128-column
    slabs, diagonal tiles only up to the diagonal (~n^2/2 + 32 n; the tiles are STORED whole,
    n^2/2 + 64 n)."""
    return 0.5 * n * n + 32.0 * n


# Eigen form: the device stores U once and, for blocks of up to EIGEN_FUSED_MAX_ROWS SNPs, streams
# it ONCE per product (ld_eig_fused_kernel: a slab of columns stays in registers between
# t' = s * (U^T x) and y = U t'); taller blocks stream it twice (two-pass kernels).  Per element the
# fused kernel runs at ~0.85 of the dense kernel's rate (workload C4, one right-hand side: 5.2-5.8
# TB/s per block-height class against 6.2 TB/s; r02q: all-eigen 432 sweeps/s, all-dense 331).
EIGEN_FORM_PENALTY = 1.2
EIGEN_FUSED_MAX_ROWS = 6144          # csrc/kernels.hip EIG_MAX_ROWS (512 threads x 12 rows)


def _eigen_passes(n):
    return 1.0 if n <= EIGEN_FUSED_MAX_ROWS else 2.0


def eigen_cost(n, r):
    """Elements a product streams for an eigen-form block, in dense-kernel elements."""
    return EIGEN_FORM_PENALTY * _eigen_passes(n) * n * r


def dense_is_cheaper(n, r):
    """Dense symmetric form (lower triangle once, ~n^2/2 + 32 n elements) vs eigen form (U once,
    n r), by the measured cost of a product: eigen form only for r < (n/2 + 32) / 1.2."""
    return dense_bytes_moved(n) <= eigen_cost(n, r)


def max_eigen_rank(n):
    """Largest rank for which `auto` keeps a block of n SNPs in eigen form (at least 1)."""
    return max(1, min(n, int(dense_bytes_moved(n) / (EIGEN_FORM_PENALTY * _eigen_passes(n) * n))))


class LowRankMatrix:
    """Symmetric block stored as u diag(s) v + diag(D), v = u^T
    (reference matrix_structures.py:38-234)."""

    _LAZY_FIELDS = ('u', 's', 'v', 'D', 'inv_s')

    @classmethod
    def deferred(cls, make_matrix, n, t=1.0):
        """A block whose dense matrix is produced (and eigendecomposed) only when first needed:
        `make_matrix()` returns the symmetric n x n matrix.  Lets a fit decompose only the
        blocks of its own shard, in parallel (BlockDiagonalMatrix.materialize)."""
        self = cls.__new__(cls)
        self._thunk = (make_matrix, t)
        self.shape = (int(n), int(n))
        return self

    def is_deferred(self):
        return self.__dict__.get('_thunk') is not None

    def materialize(self):
        thunk = self.__dict__.get('_thunk')
        if thunk is not None:
            make_matrix, t = thunk
            self._thunk = None
            self.__init__(make_matrix(), t)
            self._from_thunk = thunk
        return self

    def forget(self):
        """Drop the factors of a block that came from a thunk (they can be recomputed on
        demand): the streaming device loader calls this once a block lives in HBM, so the host
        never holds more than a window of decomposed blocks."""
        thunk = self.__dict__.get('_from_thunk')
        if thunk is not None:
            for name in LowRankMatrix._LAZY_FIELDS:
                self.__dict__.pop(name, None)
            self._from_thunk = None
            self._thunk = thunk
        return self

    def __getattr__(self, name):
        # only reached when normal lookup fails, i.e. for the factor fields of a deferred block
        if name in LowRankMatrix._LAZY_FIELDS and self.__dict__.get('_thunk') is not None:
            self.materialize()
            return self.__dict__[name]
        raise AttributeError(name)

    def __init__(self, X=None, t=1.0, u=None, s=None, v=None, D=None, hdf_file=None):
        if hdf_file is not None:
            raise NotImplementedError('--mmap/HDF5-backed LD is not supported: the MI355X '
                                      'build keeps LD resident in HBM')
        parts = (u, s, v, D)
        if X is not None:
            if any(p is not None for p in parts):
                raise ValueError('Cannot provide both a matrix and an SVD decomposition')
            X = np.asarray(X, dtype=np.float64)
            if not np.allclose(X, X.T):
                raise ValueError('Provided matrix is not symmetric')
            u, s, v = _svd_threshold(X, t)
            D = np.zeros(X.shape[0])
        else:
            if any(p is None for p in parts):
                raise ValueError('Need to provide either a matrix or an SVD decomposition')
            sel = np.flatnonzero(s >= 1 - np.sqrt(t))
            u, s, v = u[:, sel], s[sel], v[sel, :]
        self.D = np.array(D, dtype=np.float64)
        big = s > 1e-12 * np.max(s)
        if big.any():
            self.u = np.ascontiguousarray(u[:, big], dtype=np.float64)
            self.s = np.array(s[big], dtype=np.float64)
            self.v = np.ascontiguousarray(v[big, :], dtype=np.float64)
            self.inv_s = 1.0 / self.s
        else:
            self.u = np.ascontiguousarray(u[:, :1], dtype=np.float64)
            self.s = np.zeros(1)
            self.v = np.ascontiguousarray(v[:1, :], dtype=np.float64)
            self.inv_s = np.zeros(1)
        self.shape = (self.u.shape[0], self.v.shape[1])

    # -- host-side (load-time) operations ------------------------------------------------
    def dot(self, vector):
        """u @ (s * (v @ x)) + D * x for ONE block on the host.  Used only at load time by
        single-block callers; the fit loop multiplies through BlockDiagonalMatrix.dot (GPU)."""
        mid = (self.s * self.v.dot(vector).T).T
        return self.u.dot(mid) + (self.D * np.asarray(vector).T).T

    def dot_i(self, vector, i):
        return self.u[i].dot(self.s * self.v.dot(vector)) + self.D[i] * vector[i]

    def inverse_dot(self, vector):
        """PseudoInverse(block) @ vector (reference matrix_structures.py:159-196)."""
        zero_d = np.isclose(np.abs(self.D), 0)
        if zero_d.any():
            if np.isclose(self.D, 0).all():
                return self.v.T.dot(self.u.T.dot(vector) * self.inv_s)
            full = np.diag(self.D) + (self.u * self.s).dot(self.v)
            ev = np.linalg.eigh(full)[0][::-1]
            where = np.flatnonzero(np.isclose(np.cumsum(ev) / np.sum(ev), 1.))
            cut = where[0] if where.size else len(ev) - 1
            return np.linalg.pinv(full, rcond=ev[cut] / ev[0] * 0.1).dot(vector)
        core = np.linalg.inv(np.diag(self.inv_s) + self.v.dot((self.u.T / self.D).T))
        y = vector / self.D
        return y - self.u.dot(core.dot(self.v.dot(y))) / self.D

    def diag(self):
        return np.einsum('ik,ki->i', self.u * self.s, self.v) + self.D

    def matrix_power(self, power):
        if not np.allclose(self.D, 0):
            raise NotImplementedError('Matrix powers where the diagonal approximation is not '
                                      'zero have not yet been implemented.')
        return LowRankMatrix(u=self.u, s=self.s ** power, v=self.v, D=self.D)

    def get_rank(self):
        if np.allclose(self.D, 0):
            if self.s.shape[0] > 1:
                return self.s.shape[0]
            return 0 if self.s[0] == 0 else 1
        if np.all(self.D > 0):
            return self.D.shape[0]
        full = np.diag(self.D) + (self.u * self.s).dot(self.v)
        return np.linalg.matrix_rank(full, hermitian=True)

    def reconstruct(self):
        """u diag(s) u^T -- the dense matrix the GPU multiplies by when the block is kept in
        dense form (NOT the raw .npy matrix)."""
        return (self.u * self.s).dot(self.v)


class BlockDiagonalMatrix:
    """Block-diagonal symmetric operator over SNPs in `perm` order with implicit zero
    rows/columns at `missing` (reference matrix_structures.py:237-447)."""

    def __init__(self, matrices, inverse=False, perm=None, missing=None):
        self.missing = (np.array([], dtype=np.int64) if missing is None
                        else np.array(missing))
        for m in matrices:
            if not isinstance(m, LowRankMatrix):
                raise ValueError('Component matrices must be of type LowRankMatrix')
        self.matrices = matrices
        self._inverted = inverse
        self.starts = np.cumsum([0] + [m.shape[0] for m in matrices])
        n = int(self.starts[-1]) + self.missing.shape[0]
        self.shape = (n, n)
        if perm is None:
            self.perm = np.arange(n)
        else:
            if perm.shape[0] != n:
                raise ValueError('perm must be a vector conformal to the non-missing parts '
                                 'of the matrix.')
            self.perm = np.array(perm)
        self.inv_perm = np.argsort(self.perm)
        if not np.allclose(self.perm[self.inv_perm], np.arange(n)):
            raise ValueError('perm and missing should together contain all of the indices. '
                             'Some are missing.')
        self._engine = None

    def materialize(self, block_ids=None, workers=None):
        """Eigendecompose the deferred blocks among `block_ids` (default: all) on a thread pool;
        LAPACK releases the GIL, BLAS is limited to one thread per worker meanwhile."""
        todo = [self.matrices[b] for b in (range(len(self.matrices)) if block_ids is None
                                           else block_ids) if self.matrices[b].is_deferred()]
        parallel_map(lambda m: m.materialize(), todo, workers)
        return self

    # -- device operator -------------------------------------------------------------------
    def device_blocks(self, form='auto'):
        """Blocks in the form the HIP LD store takes, chosen by bytes streamed per product: the
        dense symmetric form reads the lower triangle once (~n^2/2 + 32 n elements), the eigen
        form reads U twice (dense_is_cheaper)."""
        def one(m):
            n, r = m.u.shape
            if not np.allclose(m.D, 0):
                raise NotImplementedError('device LD blocks must have a zero diagonal part')
            dense = form == 'dense' or (form == 'auto' and dense_is_cheaper(n, r))
            return ('dense', m.reconstruct()) if dense else ('eig', m.u, m.s)
        return parallel_map(one, self.matrices)

    def _own_engine(self):
        if self._engine is None:
            from .engine import HipEngine
            eng = HipEngine(1, self.shape[0], 2, 1)
            eng.load_ld(0, self.device_blocks(), self.perm.astype(np.int64),
                        int(self.starts[-1]))
            self._engine = eng
        return self._engine

    def dot(self, vector):
        """Matrix @ vector on the GPU (vilma_ld_matvec); raises if no GPU / library."""
        if self._inverted:
            return self._per_block(vector, lambda m, x, lo: m.inverse_dot(x))
        vector = np.asarray(vector, dtype=np.float64)
        if vector.ndim != 1:
            return np.stack([self.dot(col) for col in vector.T], axis=1)
        return self._own_engine().ld_matvec(vector[None, :])[0]

    # -- host-side (load-time) operations ------------------------------------------------
    def _per_block(self, vector, fn):
        x = np.asarray(vector)[self.perm]
        parts = parallel_map(lambda ml: fn(ml[0], x[ml[1]:ml[1] + ml[0].shape[0]], ml[1]),
                             list(zip(self.matrices, self.starts[:-1])))
        parts.append(np.zeros([self.missing.shape[0]] + list(x.shape[1:])))
        return np.concatenate(parts, axis=0)[self.inv_perm]

    def dot_i(self, vector, i):
        if self._inverted:
            raise NotImplementedError('dot_i with inverted matrices has not been '
                                      'implemented yet.')
        if i in self.missing:
            return 0.
        t = self.inv_perm[i]
        b = np.searchsorted(self.starts, t, 'right') - 1
        lo = self.starts[b]
        m = self.matrices[b]
        return m.dot_i(vector[self.perm][lo:lo + m.shape[0]], t - lo)

    def ridge_inverse_dot(self, vector, regularizer):
        """Inverse(Matrix + diag(regularizer)) @ vector, block by block
        (reference matrix_structures.py:349-387)."""
        if self._inverted:
            raise NotImplementedError('ridge_inverse_dot with inverted matrices has not been '
                                      'implemented yet.')
        reg = np.zeros_like(vector)
        reg[:] = regularizer
        reg = reg[self.perm]

        def solve(m, x, lo):
            shifted = LowRankMatrix(u=m.u, s=m.s, v=m.v, D=m.D + reg[lo:lo + m.shape[0]])
            return shifted.inverse_dot(x)
        return self._per_block(vector, solve)

    def matrix_power(self, power):
        return BlockDiagonalMatrix([m.matrix_power(power) for m in self.matrices],
                                   inverse=self._inverted, missing=self.missing)

    @property
    def inverse(self):
        return BlockDiagonalMatrix(self.matrices, inverse=not self._inverted, perm=self.perm,
                                   missing=self.missing)

    def diag(self):
        if self._inverted:
            raise NotImplementedError('Getting the diagonal of an inverted matrix has not '
                                      'been implemented yet.')
        parts = parallel_map(lambda m: m.diag(), self.matrices) + [np.zeros(self.missing.shape[0])]
        return np.concatenate(parts, axis=0)[self.inv_perm]

    def get_rank(self):
        return sum(m.get_rank() for m in self.matrices)
