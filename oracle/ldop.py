"""Oracle restatement of the reference's block-diagonal LD operator (matrix_structures.py).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  `EigenBlock` follows `LowRankMatrix`
(matrix_structures.py:38-234) restricted to what `vilma fit` uses (the diagonal part D is
always zero on this path except inside `ridge_inverse_dot`), `BlockDiagonalLD` follows
`BlockDiagonalMatrix` (matrix_structures.py:237-447).
"""
import numpy as np


def eig_threshold(matrix, ld_thresh):
    """matrix_structures.py:15-28 -- eigh, keep eigenvalues >= 1 - sqrt(t); if none
    survive return the rank-1 zero stand-in (ones vectors, zero value)."""
    vals, vecs = np.linalg.eigh(matrix)
    keep = np.where(vals >= 1 - np.sqrt(ld_thresh))[0]
    if len(keep) == 0:
        n = matrix.shape[0]
        return np.ones((n, 1)), np.zeros(1)
    return np.copy(vecs[:, keep]), np.copy(vals[keep])


class EigenBlock:
    """One LD block held as U diag(s) U^T (matrix_structures.py:72-146)."""

    def __init__(self, X=None, t=1.0, u=None, s=None):
        if X is not None:
            if not np.allclose(X, X.T):
                raise ValueError('Provided matrix is not symmetric')
            u, s = eig_threshold(X, t)
        else:
            sel = np.where(s >= 1 - np.sqrt(t))[0]      # :113-116
            u, s = u[:, sel], s[sel]
        big = s > (1e-12 * np.max(s))                    # :119
        if big.sum() > 0:
            self.u = np.ascontiguousarray(u[:, big])
            self.s = np.copy(s[big])
            self.inv_s = 1. / self.s
        else:                                            # :141-145
            self.u = np.ascontiguousarray(u[:, :1])
            self.s = np.zeros(1)
            self.inv_s = np.zeros(1)
        self.v = np.ascontiguousarray(self.u.T)
        self.shape = (self.u.shape[0], self.u.shape[0])

    def dot(self, vector):
        """matrix_structures.py:148-152 -- two GEMVs: u @ (s * (v @ x))."""
        return self.u.dot(self.s * self.v.dot(vector))

    def pinv_dot(self, vector):
        """matrix_structures.py:159-166 with D == 0."""
        return self.v.T.dot(self.u.T.dot(vector) * self.inv_s)

    def ridge_solve(self, vector, D):
        """inverse(U diag(s) U^T + diag(D)) @ vector -- matrix_structures.py:159-196 as
        reached from ridge_inverse_dot (:376-383), all three branches."""
        near0 = np.isclose(np.abs(D), 0)
        if np.any(near0):
            if np.all(np.isclose(D, 0)):
                return self.pinv_dot(vector)
            reconst = np.diag(D) + (self.u * self.s).dot(self.v)
            e_vals = np.linalg.eigh(reconst)[0][::-1]
            hit = np.where(np.isclose(np.cumsum(e_vals) / np.sum(e_vals), 1.))[0]
            cut = hit[0] if len(hit) > 0 else len(e_vals) - 1
            rcond = e_vals[cut] / e_vals[0] * 0.1
            return np.linalg.pinv(reconst, rcond=rcond).dot(vector)
        small = np.linalg.inv(np.diag(self.inv_s) + self.v.dot((self.u.T / D).T))
        scaled = vector / D
        return scaled - self.u.dot(small.dot(self.v.dot(scaled))) / D

    def diag(self):
        """matrix_structures.py:198-203."""
        return np.einsum('ik,ki->i', self.u * self.s, self.v)

    def get_rank(self):
        """matrix_structures.py:213-222 (D == 0 branch)."""
        if self.s.shape[0] > 1:
            return self.s.shape[0]
        return 0 if self.s[0] == 0 else 1


class BlockDiagonalLD:
    """matrix_structures.py:237-447: blocks laid out in `perm` order, `missing` SNPs are
    implicit zero rows/columns appended after the blocks."""

    def __init__(self, blocks, perm=None, missing=None):
        self.missing = (np.array([], dtype=np.int64) if missing is None
                        else np.copy(missing))
        self.blocks = list(blocks)
        self.starts = np.cumsum([0] + [b.shape[0] for b in self.blocks])
        n = int(self.starts[-1]) + self.missing.shape[0]
        self.shape = (n, n)
        self.perm = np.arange(n) if perm is None else np.copy(perm)
        if self.perm.shape[0] != n:
            raise ValueError('perm must be a vector conformal to the non-missing parts '
                             'of the matrix.')
        self.inv_perm = np.argsort(self.perm)
        if not np.array_equal(self.perm[self.inv_perm], np.arange(n)):
            raise ValueError('perm and missing should together contain all of the indices.')

    def _blockwise(self, vector, fn):
        x = vector[self.perm]
        parts = [fn(b, x[s:s + b.shape[0]], s) for b, s in zip(self.blocks, self.starts[:-1])]
        parts.append(np.zeros(self.missing.shape[0]))
        return np.concatenate(parts)[self.inv_perm]

    def dot(self, vector):
        """matrix_structures.py:389-408."""
        return self._blockwise(vector, lambda b, x, s: b.dot(x))

    def inverse_dot(self, vector):
        """matrix_structures.py:418-424 + :396-399 (lazy pseudo-inverse)."""
        return self._blockwise(vector, lambda b, x, s: b.pinv_dot(x))

    def ridge_inverse_dot(self, vector, regularizer):
        """matrix_structures.py:349-387."""
        reg = np.zeros_like(vector)
        reg[:] = regularizer
        reg = reg[self.perm]
        return self._blockwise(vector,
                               lambda b, x, s: b.ridge_solve(x, reg[s:s + b.shape[0]]))

    def diag(self):
        """matrix_structures.py:426-440."""
        parts = [b.diag() for b in self.blocks] + [np.zeros(self.missing.shape[0])]
        return np.concatenate(parts)[self.inv_perm]

    def get_rank(self):
        """matrix_structures.py:442-447."""
        return sum(b.get_rank() for b in self.blocks)
