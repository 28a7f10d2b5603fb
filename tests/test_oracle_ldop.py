"""Oracle LD operator vs vectors produced by the reference's matrix_structures.py."""
import numpy as np
import pytest

from helpers import golden
from oracle.ldop import EigenBlock, BlockDiagonalLD

K = golden('ldop_kat.npz')


@pytest.mark.parametrize('t', [1.0, 0.8, 0.3])
def test_block_diagonal_kat(t):
    tag = 't%02d_' % int(t * 10)
    blocks = [EigenBlock(K['X%d' % b], t) for b in range(3)]
    bd = BlockDiagonalLD(blocks, perm=K['perm'], missing=K['missing'])
    assert np.array_equal(bd.starts, K[tag + 'starts'])
    assert np.array_equal(bd.inv_perm, K[tag + 'inv_perm'])
    assert bd.get_rank() == int(K[tag + 'rank'])
    assert [b.get_rank() for b in blocks] == list(K[tag + 'ranks'])
    for b, blk in enumerate(blocks):
        np.testing.assert_allclose(blk.s, K[tag + 's%d' % b], rtol=1e-10)
        np.testing.assert_allclose((blk.u * blk.s) @ blk.v, K[tag + 'recon%d' % b], atol=1e-12)
    np.testing.assert_allclose(bd.dot(K['vec']), K[tag + 'dot'], atol=1e-12)
    np.testing.assert_allclose(bd.inverse_dot(K['vec']), K[tag + 'inv_dot'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(bd.ridge_inverse_dot(K['vec'], K['reg']), K[tag + 'ridge'],
                               rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(bd.ridge_inverse_dot(K['vec'], 0.7), K[tag + 'ridge_scalar'],
                               rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(bd.diag(), K[tag + 'diag'], atol=1e-12)
    assert np.all(bd.dot(K['vec'])[K['missing']] == 0)


def test_degenerate_block():
    blk = EigenBlock(K['degenerate_X'], 0.5)
    assert np.array_equal(blk.s, K['degenerate_s'])
    assert blk.get_rank() == int(K['degenerate_rank']) == 0
    assert np.array_equal(blk.dot(np.arange(4.0)), K['degenerate_dot'])
