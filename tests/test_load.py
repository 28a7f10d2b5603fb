"""Loaders and mixture grid of the product against the reference's own fixtures and vectors
recorded from the reference's load.py / vi_options.py (tests/golden/loader_kat.npz,
mixgrid_kat.npz).  Assertions mirror reference tests/test.py:486-706."""
import os

import numpy as np
import pytest

from helpers import golden, GOLDEN
from vilma_amd import load
from vilma_amd import vi_options

REF = os.path.join(GOLDEN, 'refdata')
EX = os.path.join(GOLDEN, 'example')
K = golden('loader_kat.npz')


def ref(name):
    return os.path.join(REF, name)


def test_load_variant_list():
    for bad in ('bad_variants_missing_id.tsv', 'bad_variants_missing_a1.tsv',
                'bad_variants_missing_a2.tsv'):
        with pytest.raises(ValueError):
            load.load_variant_list(ref(bad))
    variants = load.load_variant_list(ref('good_variants.tsv'))
    assert len(variants) == 13
    assert list(variants.columns) == ['ID', 'A1', 'A2']


def test_load_annotations():
    variants = load.load_variant_list(ref('good_variants.tsv'))
    null, deny = load.load_annotations(None, variants)
    assert null.shape == (13, 1) and np.allclose(null, 1) and deny == []
    ann, deny = load.load_annotations(ref('good_annotations.tsv'), variants)
    assert ann.shape == (13, 6)
    assert np.array_equal(np.asarray(ann, dtype=float), K['annotations'])
    assert deny == K['annotations_denylist'].tolist() == [12]
    with pytest.raises(ValueError):
        load.load_annotations(ref('bad_annotations_missing_id.tsv'), variants)
    with pytest.raises(ValueError):
        load.load_annotations(ref('bad_annotations_missing_annotation.tsv'), variants)


@pytest.mark.parametrize('name', ['good_sumstats_beta', 'good_sumstats_or', 'good_sumstats_flip'])
def test_load_sumstats(name):
    variants = load.load_variant_list(ref('good_variants.tsv'))
    stats, missing = load.load_sumstats(ref(name + '.tsv'), variants)
    assert len(stats) == 13
    assert missing == K[name + '_missing'].tolist()
    assert np.array_equal(np.array(stats.BETA, dtype=float), K[name + '_BETA'])
    assert np.array_equal(np.array(stats.SE, dtype=float), K[name + '_SE'])


def test_load_sumstats_errors_and_missing():
    variants = load.load_variant_list(ref('good_variants.tsv'))
    for bad in ('id', 'beta', 'se', 'a1', 'a2'):
        with pytest.raises(ValueError):
            load.load_sumstats(ref('bad_sumstats_missing_%s.tsv' % bad), variants)
    vpm = load.load_variant_list(ref('good_variants_plus_missing.tsv'))
    stats, missing = load.load_sumstats(ref('good_sumstats_beta_plus_missing.tsv'), vpm)
    assert missing == K['plusmissing_sumstats_missing'].tolist()
    assert sorted(missing) == [10, 11, 12, 14]
    assert np.array_equal(np.array(stats.BETA, dtype=float), K['plusmissing_BETA'])
    assert np.array_equal(np.array(stats.SE, dtype=float), K['plusmissing_SE'])


CASES = {
    'plain': ('ld_manifest.tsv', 'good_variants.tsv', [], 1.0),
    'thresh': ('ld_manifest.tsv', 'good_variants.tsv', [], 0.8),
    'deny': ('ld_manifest.tsv', 'good_variants.tsv', [3, 4, 5], 1.0),
    'svd': ('ld_manifest_svd.tsv', 'good_variants.tsv', [], 1.0),
    'svd_deny': ('ld_manifest_svd.tsv', 'good_variants.tsv', [3, 4, 5], 0.8),
    'plusmissing': ('ld_manifest.tsv', 'good_variants_plus_missing.tsv', [], 1.0),
}


@pytest.mark.parametrize('tag', sorted(CASES))
def test_load_ld_from_schema(tag):
    """SNP -> block assignment (perm, missing, starts) is BIT exact; block contents match."""
    manifest, varfile, deny, t = CASES[tag]
    variants = load.load_variant_list(ref(varfile))
    bd, missing = load.load_ld_from_schema(ref(manifest), variants, deny, t, False)
    assert np.array_equal(bd.perm, K[tag + '_perm']) and bd.perm.dtype == K[tag + '_perm'].dtype
    assert missing == K[tag + '_missing_list'].tolist()
    assert np.array_equal(bd.missing, K[tag + '_missing'])
    assert np.array_equal(bd.starts, K[tag + '_starts'])
    assert bd.get_rank() == int(K[tag + '_rank'])
    np.testing.assert_allclose(bd.diag(), K[tag + '_diag'], atol=1e-13)
    for b, m in enumerate(bd.matrices):
        np.testing.assert_allclose(m.reconstruct(), K[tag + '_recon%d' % b], atol=1e-13)
        np.testing.assert_allclose(m.s, K[tag + '_s%d' % b], rtol=1e-12)
    v = np.linspace(-1, 1, bd.shape[0])
    np.testing.assert_allclose(bd.inverse.dot(v), K[tag + '_invdot'], rtol=1e-9, atol=1e-12)


def test_load_missing_and_flips():
    """reference tests/test.py:594-606, 674-696: flips at rows 0,2 => R[0,2] = -1; SNPs 5 and
    12 (and 13, 14 with the extended variant list) have no LD."""
    variants = load.load_variant_list(ref('good_variants.tsv'))
    bd, missing = load.load_ld_from_schema(ref('ld_manifest.tsv'), variants, [], 1., False)
    assert sorted(missing) == [5, 12]
    dense = np.zeros((13, 13))
    n_ld = int(bd.starts[-1])
    dense[np.ix_(bd.perm[:n_ld], bd.perm[:n_ld])] = bd.matrices[0].reconstruct()
    want = np.eye(13)
    want[0, 2] = want[2, 0] = -1
    want[5, 5] = want[12, 12] = 0
    # the fixture block is singular (rows 0 and 2 are anti-correlated copies): the operator is
    # the PSD reconstruction, which reproduces the matrix itself here
    np.testing.assert_allclose(dense, want, atol=1e-12)
    vpm = load.load_variant_list(ref('good_variants_plus_missing.tsv'))
    bd, missing = load.load_ld_from_schema(ref('ld_manifest.tsv'), vpm, [], 1., False)
    assert sorted(missing) == [5, 12, 13, 14] and len(missing) == 4
    with pytest.raises(NotImplementedError):
        load.load_ld_from_schema(ref('ld_manifest.tsv'), vpm, [], 1., True)    # --mmap


def test_example_schema():
    variants = load.load_variant_list(os.path.join(EX, 'keep_variants.txt'))
    bd, missing = load.load_ld_from_schema(os.path.join(EX, 'ld_mat', 'example_schema.schema'),
                                           variants, [], 1.0, False)
    assert np.array_equal(bd.perm, K['example_perm'])
    assert missing == K['example_missing_list'].tolist()
    assert np.array_equal(bd.starts, K['example_starts'])
    for b, m in enumerate(bd.matrices):
        np.testing.assert_allclose(m.reconstruct(), K['example_recon%d' % b], atol=1e-13)


@pytest.mark.parametrize('P,Kc', [(1, 5), (2, 3), (3, 2)])
def test_mixture_grid_rng_order(P, Kc):
    """_make_simple reproduces the reference grid AND leaves the legacy RNG in the same state
    (vi_options.py:301-337); M = K+2 for P=1, 3(K+2)K^(P(P-1)/2) + 3P(K+1) otherwise."""
    G = golden('mixgrid_kat.npz')
    tag = 'P%d_K%d' % (P, Kc)
    np.random.seed(42)
    covs = vi_options._make_simple(P, Kc, G[tag + '_mins'], G[tag + '_maxes'])
    assert np.array_equal(np.array(covs), G[tag])
    assert np.random.uniform() == float(G[tag + '_next_uniform'])
    want_m = Kc + 2 if P == 1 else 3 * (Kc + 2) * Kc ** (P * (P - 1) // 2) + 3 * P * (Kc + 1)
    assert len(covs) == want_m


def test_lazy_loading_is_equivalent_and_deferred():
    """lazy=True defers np.load + eigh per block until first use; assignment and factors equal."""
    variants = load.load_variant_list(ref('good_variants.tsv'))
    eager, miss_e = load.load_ld_from_schema(ref('ld_manifest.tsv'), variants, [3, 4], 0.8)
    lazy, miss_l = load.load_ld_from_schema(ref('ld_manifest.tsv'), variants, [3, 4], 0.8, lazy=True)
    assert miss_e == miss_l and np.array_equal(eager.perm, lazy.perm)
    assert np.array_equal(eager.starts, lazy.starts) and lazy.shape == eager.shape
    assert all(m.is_deferred() for m in lazy.matrices)
    lazy.materialize(workers=2)
    assert not any(m.is_deferred() for m in lazy.matrices)
    for a, b in zip(eager.matrices, lazy.matrices):
        np.testing.assert_allclose(b.reconstruct(), a.reconstruct(), atol=1e-14)
        assert b.get_rank() == a.get_rank()
    variants = load.load_variant_list(os.path.join(EX, 'keep_variants.txt'))
    lazy, _ = load.load_ld_from_schema(os.path.join(EX, 'ld_mat', 'example_schema.schema'),
                                       variants, [], 1.0, lazy=True)
    assert lazy.matrices[1].is_deferred()
    _ = lazy.matrices[1].s                      # touching a factor materialises just that block
    assert not lazy.matrices[1].is_deferred() and lazy.matrices[0].is_deferred()


def test_store_upper_bound_covers_every_rank():
    from vilma_amd import _lib, ld_device
    from vilma_amd.matrix_structures import dense_is_cheaper
    lib = _lib.load()
    for n in (1, 2, 15, 16, 17, 20, 33, 127, 128, 129, 300, 1000, 2431):
        for form in ('auto', 'dense', 'eig'):
            bound = ld_device.store_upper_bound(lib, [n], form)
            for r in sorted({min(n, v) for v in (1, 2, n // 7 + 1, n // 3 + 1, n // 2 + 1, n)}):
                dense = form == 'dense' or (form == 'auto' and dense_is_cheaper(n, r))
                need = lib.vilma_ld_dense_elems(n) if dense else lib.vilma_ld_lowrank_elems(n, r)
                assert need <= bound, (n, r, form, need, bound)


def test_select_eigenpairs_is_lowrankmatrix_selection():
    """matrix_structures.select_eigenpairs (used when eigh runs on the GPU) == what
    LowRankMatrix(X, t) keeps (reference matrix_structures.py:15-28, 136-146)."""
    from vilma_amd.matrix_structures import LowRankMatrix, select_eigenpairs
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.normal(size=(9, 9)))
    spectra = [np.linspace(0.01, 2, 9), np.r_[np.zeros(4), 1e-14, 1e-13, 0.2, 1, 3],
               -np.ones(9), np.zeros(9), np.r_[-0.5, 0, 0, 1e-20, 1e-15, 1e-3, 0.1, 0.5, 4]]
    for w in spectra:
        X = (q * w) @ q.T
        X = 0.5 * (X + X.T)
        for t in (1.0, 0.8, 0.3):
            host = LowRankMatrix(X, t)
            ev = np.linalg.eigh(X)[0]
            idx, degenerate = select_eigenpairs(ev, t)
            if degenerate == 'ones':
                assert host.s.shape == (1,) and host.s[0] == 0 and np.all(host.u == 1)
            elif degenerate == 'zero':
                assert host.s.shape == (1,) and host.s[0] == 0
            else:
                np.testing.assert_array_equal(ev[idx], host.s)


def test_gpu_eigh_plan_and_stacks(monkeypatch):
    """The loader's cost model (ld_device.plan_gpu_eigh / gpu_stacks): all but the small blocks go
    to the GPU, in stacks padded to their largest member, none smaller than half of that, no
    stack beyond the memory cap; nothing without a deferred matrix goes."""
    from vilma_amd import ld_device
    from vilma_amd.synthetic import block_sizes
    sizes = [int(n) for n in block_sizes(1_000_000, 1700, None, 0)]
    chosen = ld_device.plan_gpu_eigh(sizes, [True] * len(sizes), 16)
    assert all(sizes[b] > 128 for b in chosen) and all(sizes[b] < 320 for b in range(len(sizes))
                                                       if b not in chosen)
    assert len(chosen) > 0.9 * len(sizes)
    stacks = ld_device.gpu_stacks(sizes, chosen)
    assert sorted(b for st in stacks for b in st) == sorted(chosen)
    for st in stacks:
        npad = ld_device._padded(max(sizes[b] for b in st))
        assert all(2 * sizes[b] >= npad for b in st)
        assert len(st) * 8 * npad * npad <= ld_device.EIGH_STACK_BYTES or len(st) == 1
        assert len(st) <= ld_device.EIGH_STACK_MAX
    assert len(stacks) < len(chosen) / 8                 # real stacks, not singletons
    assert ld_device.plan_gpu_eigh(sizes, [False] * len(sizes), 16) == set()
    monkeypatch.setenv('VILMA_GPU_EIGH', '0')
    assert ld_device.plan_gpu_eigh(sizes, [True] * len(sizes), 16) == set()


@pytest.mark.parametrize('body', [
    'rs1 1 100 0.0 A C\nrs2 1 200 0.0 G T\n',                        # plain
    'rs1\t1\t100\t0.0\tA\tC\n\n  rs2 1 200 0.0 G T  \n',             # tabs, a blank line, padding
    '1:100_A_C 1 100 0.0 A C\n2 1 200 0.0 G T\n',                    # one numeric-looking ID among strings
    '101 1 100 0.0 A C\n202 1 200 0.0 G T\n',                        # all-numeric IDs (pandas: int64)
    'rs1 1 100 0.0 NA C\nrs2 1 200 0.0 G T\n',                       # an NA spelling (pandas: NaN)
    'NA 1 100 0.0 A C\nrs2 1 200 0.0 G T\n',
    '"rs 1" 1 100 0.0 A C\nrs2 1 200 0.0 G T\n',                     # quotes
    'rs1 1 100 0.0 TRUE FALSE\nrs2 1 200 0.0 TRUE TRUE\n',           # boolean-looking alleles
    'rs1 1 100 0.0 A\nrs2 1 200 0.0 G T\n',                          # a short line (pandas: NaN)
])
def test_var_files_read_as_pandas_reads_them(tmp_path, body):
    """load._read_var_file splits plain .var files itself; whatever pandas.read_csv would treat
    specially must come out exactly as pandas gives it (reference load.py:262-263), since the
    SNP -> block assignment is decided by equality of these values."""
    import pandas as pd
    from vilma_amd import load
    path = tmp_path / 'block.var'
    path.write_text(body)
    want = pd.read_csv(path, header=None, sep=r'\s+', names=['ID', 'CHROM', 'BP', 'CM', 'A1', 'A2'])
    got = load._read_var_file(path)
    for have, col in zip(got, ('ID', 'A1', 'A2')):
        ref = want[col].to_numpy()
        assert len(have) == len(ref)
        for x, y in zip(have, ref):
            assert (x == y and type(x) is type(y)) or (x != x and y != y), (col, x, y)
