"""vilma_amd.matrix_structures, host side: everything the product's LowRankMatrix /
BlockDiagonalMatrix compute with numpy at load time, against (i) vectors produced by the
reference's own classes (tests/golden/ldop_kat.npz, written by make_golden.py from
/root/reference/src/vilma/matrix_structures.py) and (ii) the dense-matrix identities the
reference's tests assert (/root/reference/tests/test.py:28-477).  BlockDiagonalMatrix.dot itself
runs on the GPU: tests/test_gpu_reference_kats.py."""
import numpy as np
import pytest

from helpers import golden
from vilma_amd.matrix_structures import (LowRankMatrix, BlockDiagonalMatrix, _svd_threshold)

K = golden('ldop_kat.npz')


def _spd(rng, n):
    x = rng.random((n, n))
    return x + x.T + 3 * np.eye(n)


def test_svd_threshold_keeps_what_the_reference_keeps():
    rng = np.random.default_rng(0)
    x = _spd(rng, 5)
    u, s, v = _svd_threshold(x, 1)
    np.testing.assert_allclose(np.einsum('ik,k,kj->ij', u, s, v), x, atol=1e-12)
    for t in np.linspace(0, 1):
        assert np.all(_svd_threshold(x, t)[1] > 1 - np.sqrt(t))
    x = np.eye(5)
    x[0, 0] = 0
    u, s, v = _svd_threshold(x, 0.5)
    assert s.shape[0] == 4
    np.testing.assert_allclose(np.einsum('ik,k,kj->ij', u, s, v), x, atol=1e-12)


def test_low_rank_matrix_construction():
    with pytest.raises(ValueError):
        x = np.eye(5)
        x[0, 1] = 2                                  # not symmetric
        LowRankMatrix(X=x)
    with pytest.raises(ValueError):
        LowRankMatrix(X=np.eye(5), u=3)              # a matrix AND factors
    with pytest.raises(ValueError):
        LowRankMatrix()
    rng = np.random.default_rng(1)
    x = _spd(rng, 5)
    m = LowRankMatrix(X=x, t=1.)
    np.testing.assert_allclose(np.einsum('ik,k,kj->ij', m.u, m.s, m.v), x, atol=1e-12)
    assert m.shape == (5, 5) and np.allclose(m.inv_s, 1. / m.s) and np.allclose(m.D, 0)
    u, s, v = np.linalg.svd(x)
    m = LowRankMatrix(u=u, s=s, v=v, D=np.zeros(5))
    assert np.allclose(m.u, u) and np.allclose(m.v, v) and np.allclose(m.s, s)
    x = np.eye(5)
    x[0, 0] = 0
    m = LowRankMatrix(X=x)
    assert m.shape == (5, 5) and m.s.shape[0] == 4 and m.u.shape == (5, 4) and m.v.shape == (4, 5)


def test_low_rank_matrix_dot_dot_i_diag_with_and_without_a_diagonal_part():
    rng = np.random.default_rng(2)
    x = _spd(rng, 6)
    m = LowRankMatrix(X=x, t=1.)
    v = rng.random(6)
    np.testing.assert_allclose(m.dot(v), x.dot(v), rtol=1e-12)
    np.testing.assert_allclose(m.diag(), np.diag(x), rtol=1e-12)
    for i in range(6):
        np.testing.assert_allclose(m.dot_i(v, i), x[i].dot(v), rtol=1e-12)
    D = rng.random(6)
    md = LowRankMatrix(u=m.u, s=m.s, v=m.v, D=D)
    np.testing.assert_allclose(md.dot(v), (x + np.diag(D)).dot(v), rtol=1e-12)
    np.testing.assert_allclose(md.diag(), np.diag(x) + D, rtol=1e-12)
    for i in range(6):
        np.testing.assert_allclose(md.dot_i(v, i), (x + np.diag(D))[i].dot(v), rtol=1e-12)
    # a matrix of right-hand sides
    V = rng.random((6, 3))
    np.testing.assert_allclose(md.dot(V), (x + np.diag(D)).dot(V), rtol=1e-12)


def test_low_rank_matrix_inverse_dot_three_regimes():
    rng = np.random.default_rng(3)
    x = _spd(rng, 7)
    v = rng.random(7)
    m = LowRankMatrix(X=x, t=1.)
    np.testing.assert_allclose(m.inverse_dot(v), np.linalg.solve(x, v), rtol=1e-9)       # D = 0
    D = rng.random(7) + 0.1
    md = LowRankMatrix(u=m.u, s=m.s, v=m.v, D=D)                                            # D > 0: Woodbury
    np.testing.assert_allclose(md.inverse_dot(v), np.linalg.solve(x + np.diag(D), v), rtol=1e-9)
    D0 = D.copy()
    D0[:3] = 0                                                                              # mixed: pinv
    mm = LowRankMatrix(u=m.u, s=m.s, v=m.v, D=D0)
    np.testing.assert_allclose(mm.inverse_dot(v), np.linalg.solve(x + np.diag(D0), v), rtol=1e-7)
    # rank deficient, D = 0: the pseudo-inverse
    x[:, 3] = x[:, 2]; x[3, :] = x[2, :]; x[3, 3] = x[2, 2]
    ms = LowRankMatrix(X=x, t=1.)
    np.testing.assert_allclose(ms.inverse_dot(v), np.linalg.pinv(x).dot(v), rtol=1e-6, atol=1e-9)


def test_low_rank_matrix_power_and_rank():
    rng = np.random.default_rng(4)
    x = _spd(rng, 5)
    m = LowRankMatrix(X=x, t=1.)
    half = m.matrix_power(0.5)
    np.testing.assert_allclose(half.dot(half.dot(np.eye(5))), x, rtol=1e-10)
    np.testing.assert_allclose(m.matrix_power(-1).dot(np.eye(5)), np.linalg.inv(x), rtol=1e-9)
    with pytest.raises(NotImplementedError):
        LowRankMatrix(u=m.u, s=m.s, v=m.v, D=np.ones(5)).matrix_power(2)
    assert m.get_rank() == 5
    assert LowRankMatrix(u=m.u, s=m.s, v=m.v, D=np.ones(5)).get_rank() == 5
    y = np.eye(5)
    y[0, 0] = 0
    assert LowRankMatrix(X=y).get_rank() == 4
    D = np.zeros(5)
    D[0] = 1.                                         # the diagonal part restores the missing rank
    my = LowRankMatrix(X=y)
    assert LowRankMatrix(u=my.u, s=my.s, v=my.v, D=D).get_rank() == 5
    assert LowRankMatrix(X=0.01 * np.eye(4), t=0.5).get_rank() == 0      # all below the threshold


@pytest.mark.parametrize('t', [1.0, 0.8, 0.3])
def test_block_diagonal_host_operations_against_the_reference_vectors(t):
    tag = 't%02d_' % int(t * 10)
    blocks = [LowRankMatrix(K['X%d' % b], t) for b in range(3)]
    bd = BlockDiagonalMatrix(blocks, perm=K['perm'], missing=K['missing'])
    vec, N = K['vec'], len(K['vec'])
    assert np.array_equal(bd.starts, K[tag + 'starts'])
    assert np.array_equal(bd.inv_perm, K[tag + 'inv_perm'])
    assert bd.shape == (N, N)
    assert bd.get_rank() == int(K[tag + 'rank'])
    assert [b.get_rank() for b in blocks] == list(K[tag + 'ranks'])
    for b, blk in enumerate(blocks):
        np.testing.assert_allclose(blk.s, K[tag + 's%d' % b], rtol=1e-10)
        np.testing.assert_allclose(blk.reconstruct(), K[tag + 'recon%d' % b], atol=1e-12)
    np.testing.assert_allclose(bd.diag(), K[tag + 'diag'], atol=1e-12)
    np.testing.assert_allclose([bd.dot_i(vec, i) for i in range(N)], K[tag + 'dot_i'], atol=1e-12)
    assert all(bd.dot_i(vec, int(i)) == 0 for i in K['missing'])
    np.testing.assert_allclose(bd.inverse.dot(vec), K[tag + 'inv_dot'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(bd.ridge_inverse_dot(vec, K['reg']), K[tag + 'ridge'],
                               rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(bd.ridge_inverse_dot(vec, 0.7), K[tag + 'ridge_scalar'],
                               rtol=1e-9, atol=1e-11)
    with pytest.raises(NotImplementedError):
        bd.inverse.diag()
    with pytest.raises(NotImplementedError):
        bd.inverse.dot_i(vec, 0)
    with pytest.raises(NotImplementedError):
        bd.inverse.ridge_inverse_dot(vec, 1.0)


def test_block_diagonal_construction_errors():
    rng = np.random.default_rng(5)
    a, b = LowRankMatrix(X=_spd(rng, 3)), LowRankMatrix(X=_spd(rng, 4))
    with pytest.raises(ValueError):
        BlockDiagonalMatrix([a, np.eye(4)])                       # not a LowRankMatrix
    with pytest.raises(ValueError):
        BlockDiagonalMatrix([a, b], perm=np.arange(6))            # perm of the wrong length
    with pytest.raises(ValueError):
        BlockDiagonalMatrix([a, b], perm=np.array([0, 1, 2, 3, 4, 5, 5]))     # not a permutation
    bd = BlockDiagonalMatrix([a, b], missing=np.array([7, 8]), perm=np.arange(9)[::-1].copy())
    assert bd.shape == (9, 9) and list(bd.starts) == [0, 3, 7]


def test_degenerate_block_is_a_zero_operator():
    blk = LowRankMatrix(K['degenerate_X'], 0.5)
    assert np.array_equal(blk.s, K['degenerate_s'])
    assert blk.get_rank() == int(K['degenerate_rank']) == 0
    assert np.array_equal(blk.dot(np.arange(4.0)), K['degenerate_dot'])
