"""Input readers for `vilma fit`: variant list, annotations, summary statistics and the LD
schema (manifest of .var / .npy pairs).

File formats and the SNP -> LD-block assignment follow /root/reference/src/vilma/load.py
(cited per function); the results -- `perm`, `missing`, allele-flip signs, the per-block
matrices handed to LowRankMatrix -- are bit-identical to the reference's, which is what the
parity tests pin (tests/test_load.py against the reference's own fixtures).
"""
import logging
from pathlib import Path

import numpy as np
import pandas as pd

from .matrix_structures import LowRankMatrix, BlockDiagonalMatrix

_WS = r'\s+'
_VAR_COLUMNS = ['ID', 'CHROM', 'BP', 'CM', 'A1', 'A2']


def _read_table(path, **kw):
    return pd.read_csv(path, header=0, sep=_WS, **kw)


def _fill_a2_from_ref_alt(frame):
    """A2 is whichever of REF/ALT is not A1 (reference load.py:33-35, 113-116)."""
    frame['A2'] = frame['REF'].copy()
    is_ref = frame['A1'] == frame['REF']
    frame.loc[is_ref, 'A2'] = frame.loc[is_ref, 'ALT'].copy()


def load_variant_list(variant_filename):
    """ID/A1/A2 of the SNPs to analyse, duplicates dropped, in file order = SNP order of the
    whole fit (reference load.py:21-39)."""
    variants = _read_table(variant_filename).drop_duplicates(ignore_index=True)
    for col in ('ID', 'A1'):
        if col not in variants.columns:
            raise ValueError('Variant file must contain a column labeled %s' % col)
    if 'A2' not in variants.columns:
        if 'REF' not in variants.columns or 'ALT' not in variants.columns:
            raise ValueError('Variant file must contain a column labeled A2')
        _fill_a2_from_ref_alt(variants)
    return variants[['ID', 'A1', 'A2']]


def load_annotations(annotations_filename, variants):
    """One-hot [N,A] annotation matrix and the list of SNPs without one (which get the first
    annotation) -- reference load.py:42-68.  No file -> a single all-ones column."""
    if not annotations_filename:
        return np.ones((variants.shape[0], 1)), []
    frame = _read_table(annotations_filename)
    if 'ID' not in frame.columns:
        raise ValueError('Annotation file must contain a column labeled ID')
    if 'ANNOTATION' not in frame.columns:
        raise ValueError('Annotation file must contain a column labeled ANNOTATION')
    merged = pd.DataFrame(pd.merge(variants, frame, on='ID', how='left')['ANNOTATION'])
    absent = merged['ANNOTATION'].isna()
    if absent.sum() > 0:
        logging.warning('%d out of %d total variants are missing annotations. These will get '
                        'set to having the first annotation!', absent.sum(), merged.shape[0])
    denylist = np.where(absent)[0].tolist()
    merged.loc[absent, 'ANNOTATION'] = 0
    return pd.get_dummies(merged['ANNOTATION'], dummy_na=False).to_numpy(), denylist


def load_sumstats(sumstats_filename, variants):
    """GWAS effects aligned to `variants` (reference load.py:71-139): BETA (or log OR) and SE,
    sign-flipped where the alleles are swapped; SNPs that are absent, have NaNs or whose
    alleles match neither way get BETA=0, SE=1 and are returned as missing."""
    header = pd.read_csv(sumstats_filename, nrows=1, header=0, sep=_WS)
    have = set(header.columns)
    if 'ID' not in have:
        raise ValueError('Summary Statistics File must contain a column labeled ID')
    if 'A1' not in have:
        raise ValueError('Summary Statistics File must contain a column labeled A1')
    allele_cols = ['A2']
    if 'A2' not in have:
        allele_cols = ['REF', 'ALT']
        if 'REF' not in have or 'ALT' not in have:
            raise ValueError('If summary statistics file does not contain a column labeled '
                             'A2, then it must contain REF and ALT columns.')
    if 'SE' not in have:
        raise ValueError('Summary Statistics File must contain a column labeled SE')
    effect = 'BETA' if 'BETA' in have else 'OR'
    if effect not in have:
        raise ValueError('Summary stat file needs to contain eitherBETA or OR filed.')

    table = _read_table(sumstats_filename, usecols=['ID', 'A1', 'SE', effect] + allele_cols)
    table = table[table.ID.isin(variants.ID)].reset_index(drop=True)
    if 'A2' not in table.columns:
        _fill_a2_from_ref_alt(table)
    if 'BETA' not in table.columns:
        table['BETA'] = np.log(table.OR)

    table = pd.merge(variants, table, on='ID', how='left')
    same = (table.A1_x == table.A1_y) & (table.A2_x == table.A2_y)
    swapped = (table.A1_x == table.A2_y) & (table.A1_y == table.A2_x)
    missing = table.BETA.isna() | table.SE.isna() | ((~same) & (~swapped))
    logging.warning('%d out of %d total variants are missing sumstats', missing.sum(),
                    table.shape[0])
    logging.warning('%d alleles have been flipped', swapped.sum())
    table.loc[missing, 'BETA'] = 0.
    table.loc[missing, 'SE'] = 1.
    table.loc[swapped, 'BETA'] = -table.loc[swapped, 'BETA']
    return table, np.where(missing)[0].tolist()


def schema_iterator(schema_path):
    """Yield (.var path, .npy path) per manifest line, relative to the manifest's directory
    (reference load.py:142-163)."""
    schema_path = Path(schema_path)
    base = schema_path.parents[0]
    with open(schema_path, 'r') as manifest:
        for line in manifest:
            var_name, npy_name = line.split()
            yield Path(base, var_name), Path(base, npy_name)


def load_ld_mat(ld_path, variant_indices=None, mismatch=None, signs=None):
    """One block of the schema as a dense matrix restricted to the wanted SNPs with allele
    signs applied (reference load.py:166-234).  A square .npy is the LD matrix itself; a
    (n+1) x r .npy is eigenvectors stacked on a last row of eigenvalues."""
    stored = np.load(ld_path)
    if not np.allclose(signs ** 2, 1):
        raise ValueError('signs must be a vector consisting entirely of +1s and -1s.')
    if stored.ndim == 0:
        return stored[None, None]
    rows, cols = stored.shape
    n = rows - 1 if rows > cols else rows
    if variant_indices is None:
        variant_indices = np.ones(n, dtype=bool)
    if mismatch is None:
        mismatch = np.zeros(variant_indices.sum(), dtype=bool)
    if signs is None:
        signs = np.ones(n)
    keep = ~mismatch
    if rows == cols:
        sub = np.copy(stored[np.ix_(variant_indices, variant_indices)])
        sub = sub * np.outer(signs, signs)
        return sub[np.ix_(keep, keep)]
    if rows < cols:
        raise ValueError('Bad LD matrix.')
    if rows - 1 != variant_indices.shape[0]:
        raise ValueError('Bad LD matrix.')
    vecs = np.copy(stored[:rows - 1])[variant_indices, :]
    vals = np.copy(stored[rows - 1])
    vecs = np.copy((signs.reshape((-1, 1)) * vecs)[keep])
    return (vecs * vals).dot(vecs.T)


# what pandas' C parser turns into NaN / bool in a column of strings (pandas.read_csv defaults)
_PANDAS_NA = frozenset(['', '#N/A', '#N/A N/A', '#NA', '-1.#IND', '-1.#QNAN', '-NaN', '-nan', '1.#IND',
                        '1.#QNAN', '<NA>', 'N/A', 'NA', 'NULL', 'NaN', 'None', 'n/a', 'nan', 'null'])
_PANDAS_BOOL = frozenset(['True', 'TRUE', 'true', 'False', 'FALSE', 'false'])


def _looks_numeric(token):
    try:
        float(token)
        return True
    except ValueError:
        return False


def _read_var_file(var_path):
    """(ID, A1, A2) object arrays of a block's .var file, as
    `pd.read_csv(path, header=None, sep=r'\s+', names=[ID, CHROM, BP, CM, A1, A2])` yields them
    (reference load.py:262-263).  Plain files -- six whitespace-separated fields per line, IDs and
    alleles that pandas would leave as the strings they are -- are split directly (0.1 ms instead
    of 1 ms per block: 3.4 s of a 1 M SNP x 2 cohort load are 3 400 read_csv calls); anything pandas
    would treat specially (quotes, comments, NA spellings, numeric or boolean-looking columns,
    ragged lines) goes through pandas itself."""
    with open(var_path, 'r') as handle:
        text = handle.read()
    plain = '"' not in text and "'" not in text and '\\' not in text
    if plain:
        lines = [ln.split() for ln in text.splitlines()]
        lines = [t for t in lines if t]              # (pandas skips blank lines)
        plain = bool(lines) and all(len(t) == len(_VAR_COLUMNS) for t in lines)
    if plain:
        cols = list(zip(*lines))
        ids, a1, a2 = cols[0], cols[4], cols[5]
        for col in (ids, a1, a2):
            if (any(t in _PANDAS_NA for t in col) or all(_looks_numeric(t) for t in col)
                    or all(t in _PANDAS_BOOL for t in col)):
                plain = False
                break
    if not plain:
        meta = pd.read_csv(var_path, header=None, sep=_WS, names=_VAR_COLUMNS)
        return meta['ID'].to_numpy(), meta['A1'].to_numpy(), meta['A2'].to_numpy()
    return (np.array(ids, dtype=object), np.array(a1, dtype=object), np.array(a2, dtype=object))


def _allele_match(want_a1, want_a2, have_a1, have_a2):
    """Element-wise (same alleles, swapped alleles) of two allele codings."""
    want_a1, want_a2 = np.asarray(want_a1, dtype=object), np.asarray(want_a2, dtype=object)
    have_a1, have_a2 = np.asarray(have_a1, dtype=object), np.asarray(have_a2, dtype=object)
    same = np.asarray((want_a1 == have_a1) & (want_a2 == have_a2), dtype=bool)
    swapped = np.asarray((want_a1 == have_a2) & (want_a2 == have_a1), dtype=bool)
    return same, swapped


def load_ld_from_schema(schema_path, variants, denylist, ldthresh, mmap=False, lazy=False):
    """Block-diagonal LD for `variants` from a schema (reference load.py:237-354).

    Returns (BlockDiagonalMatrix, list of SNP positions without LD).  `perm` lists, block by
    block in manifest order, the positions (in `variants`) of the SNPs each block keeps,
    followed by the SNPs no block covers.  With lazy=True the .npy files are read and
    eigendecomposed only when a block's factors are first used (so a multi-GPU fit touches only
    its own shard's blocks, in parallel); the SNP -> block assignment is unaffected.

    The reference looks every block's IDs up with `Series.isin(variants.ID)`, which hashes all
    of `variants` once per block (O(blocks x SNPs): minutes at 1 M SNPs x 1 700 blocks); here
    the ID -> position index and the denylist mask are built once.  Same assignment, bit for bit
    (tests/test_load.py against the reference's outputs)."""
    if mmap:
        raise NotImplementedError('--mmap is not supported: LD is kept resident in HBM')
    n_variants = variants.shape[0]
    id_index = pd.Index(variants['ID'])
    unique_ids = id_index.is_unique
    if not unique_ids:                      # the reference's own (slow) lookups handle repeats
        by_id = variants.set_index('ID')
        by_id['old_idx'] = np.arange(n_variants)
    denied = np.zeros(n_variants, dtype=bool)
    denied[np.asarray(denylist, dtype=np.int64)] = True
    a1_all, a2_all = variants['A1'].to_numpy(), variants['A2'].to_numpy()
    blocks, perm_parts = [], []
    n_flipped = 0
    for var_path, npy_path in schema_iterator(schema_path):
        meta_id, meta_a1, meta_a2 = _read_var_file(var_path)
        logging.info('LD matrix shape: %s', ((len(meta_id), len(meta_id)),))
        if unique_ids:
            where = id_index.get_indexer(meta_id)
            wanted = where >= 0
            if not wanted.any():
                continue
            pos = where[wanted]
        else:
            wanted = pd.Series(meta_id).isin(variants.ID).to_numpy()
            if np.sum(wanted) == 0:
                continue
            pos = by_id.loc[meta_id[wanted]].old_idx.to_numpy().flatten()
        allowed = ~denied[pos]
        wanted[np.where(wanted)[0][~allowed]] = False
        logging.info('Proportion of variant indices being used: %e', np.mean(wanted))
        pos = pos[allowed]
        if len(pos) == 0:
            continue
        same, swapped = _allele_match(a1_all[pos], a2_all[pos],
                                      meta_a1[wanted], meta_a2[wanted])
        n_flipped += swapped.sum()
        mismatch = np.logical_and(~swapped, ~same)
        if len(pos[~mismatch]) == 0:
            continue
        signs = np.ones(len(pos))
        signs[swapped] = -1
        perm_parts.append(pos[~mismatch])
        if lazy:
            def make(npy_path=npy_path, wanted=wanted.copy(), mismatch=mismatch.copy(),
                     signs=signs.copy()):
                return load_ld_mat(npy_path, wanted, mismatch, signs)
            blocks.append(LowRankMatrix.deferred(make, int((~mismatch).sum()), ldthresh))
        else:
            blocks.append(LowRankMatrix(load_ld_mat(npy_path, wanted, mismatch, signs), ldthresh))

    perm = np.concatenate(perm_parts) if perm_parts else np.array([], dtype=float)
    uncovered = set(np.arange(variants.shape[0]).tolist()) - set(perm.tolist())
    list_of_missing = list(uncovered)
    missing = np.array(list(uncovered), dtype=int)
    logging.info('Loaded a total of %d variants.', variants.shape[0])
    logging.warning('Missing LD info for %d variants. They will be ignored during '
                    'optimization.', len(missing))
    logging.warning('The alleles did not match for %d variants. They were flipped', n_flipped)
    perm = np.concatenate([perm, missing])
    if not np.all(perm == np.arange(len(perm))):
        logging.warning('The variants in the extract file and the variants in the LD matrix '
                        'were not in the same order.  The variants in the LD matrix have been '
                        'reordered to match the extract file.')
    perm = np.array(perm)
    return BlockDiagonalMatrix(blocks, perm=perm, missing=missing), list_of_missing
