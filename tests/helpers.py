"""Shared test helpers: golden loading and oracle construction (tests may import oracle/)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TRAJ_NAMES = ['p1_dense', 'p2_lowrank', 'p2_scale_se', 'p4_general', 'p1_scaled']


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def traj_blocks(g):
    """Dense per-cohort block matrices stored in a trajectory golden."""
    P = int(g['P'])
    nb = len(g['sizes'])
    return [[g['ld_%d_%d' % (p, b)] for b in range(nb)] for p in range(P)]


def oracle_from_traj(g, num_its=None):
    from oracle.ldop import EigenBlock, BlockDiagonalLD
    from oracle.vi import MultiPopVIOracle
    t = float(g['ldthresh'])
    ld = [BlockDiagonalLD([EigenBlock(X, t) for X in blocks], perm=g['perm'],
                          missing=g['missing'])
          for blocks in traj_blocks(g)]
    vi = MultiPopVIOracle(marginal_effects=g['betahat'], std_errs=g['se'], ld_mats=ld,
                          mixture_covs=list(g['covs']), annotations=g['annotations'],
                          checkpoint=False, checkpoint_freq=-1, output='t',
                          scaled=bool(g['scaled']), scale_se=bool(g['scale_se']),
                          gwas_N=g['gwas_N'], init_hg=g['init_hg'],
                          num_its=len(g['elbo']) if num_its is None else num_its)
    return vi, ld


def relerr(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300))) if a.size else 0.0
