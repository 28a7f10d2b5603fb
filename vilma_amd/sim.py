"""`vilma sim`: draw GWAS summary statistics from the mixture-of-Gaussians model, using the
GPU LD operator for the two products per cohort.

Mirror of /root/reference/src/vilma/sim.py (the adjacent caller of BlockDiagonalMatrix.dot that
SURVEY.md 8f lists as N4): same flags, same output files, and -- because the legacy numpy
RandomState is consumed in the same order (one uniform per SNP for the component, one N(0,1)
per SNP and cohort for the effect, one per SNP for the sampling noise) -- the same numbers for
a given --seed.  The per-SNP Python loops of the reference are replaced by whole-array draws
that consume the generator identically.
"""
import logging
import pickle

import numpy as np
import pandas as pd

from . import load


def args(super_parser):
    """Command line of `vilma sim` (reference sim.py:11-67)."""
    parser = super_parser.add_parser(
        'sim',
        description='Simulate GWAS summary data from a mixture-of-gaussians model.',
        usage='vilma sim <options>')
    add = parser.add_argument
    add('--sumstats', required=True, type=str,
        help='Comma-separated paths to summary statistics.')
    add('--covariance', required=True, type=str,
        help='Path to .pkl file containing the covariance matrices for each Gaussian '
             'component.')
    add('--weights', required=True, type=str,
        help='Path to a .npy file containing a matrix of weights (num_annotations x '
             'num_components) to assign to each mixture component with covariances specified '
             'by --covariance. Alternatively, can be a .npz file containing a fitted vilma '
             'model.')
    add('--gwas-n-scaling', required=False, type=str, default='1.',
        help='Comma-separated list of values to use to scale the sample sizes for each '
             'cohort.  E.g., --gwas-n-scaling 2,2 will simulate data for two cohorts with GWAS '
             'sample sizes 2x larger than the sample sizes used to generate the sumstats files '
             'provided to --sumstats.')
    add('--annotations', type=str, default='', help='Path to annotations file.')
    add('--output', required=True, type=str, help='Output path prefix.')
    add('--names', type=str, required=False,
        help='Comma-separated names of the populations for the output. Defaults to 0, 1, ...')
    add('--ld-schema', required=True, type=str,
        help='Comma-separated paths to LD panel schemas.')
    add('--seed', type=int, default=42, help='Seed for random number generation.')
    return parser


def _component_index(annotations, weights):
    """Mixture component of every SNP: SNP i gets component j with probability
    weights[annotation of i, j].  One uniform per SNP, inverted through the annotation's
    normalised cumulative weights -- exactly what RandomState.choice(p=...) does per call, so
    the generator ends up in the same state as after the reference's per-SNP loop."""
    annotations = np.asarray(annotations)
    weights = np.asarray(weights, dtype=np.float64)
    which = np.argmax(annotations == 1, axis=1)
    if not np.all(annotations[np.arange(annotations.shape[0]), which] == 1):
        raise IndexError('every SNP needs an annotation')
    tol = np.sqrt(np.finfo(np.float64).eps)
    uniforms = np.random.random_sample(annotations.shape[0])
    comp = np.zeros(annotations.shape[0], dtype=np.int64)
    for a in np.unique(which):
        p = weights[a]
        if np.any(p < 0):
            raise ValueError('probabilities are not non-negative')
        if abs(p.sum() - 1.) > tol:
            raise ValueError('probabilities do not sum to 1')
        cdf = p.cumsum()
        cdf /= cdf[-1]
        rows = which == a
        comp[rows] = cdf.searchsorted(uniforms[rows], side='right')
    return comp


def sim_components(annotations, weights):
    """One-hot [num_snps, num_components] matrix of the drawn components
    (reference sim.py:70-93)."""
    comp = _component_index(annotations, weights)
    onehot = np.zeros((comp.shape[0], np.asarray(weights).shape[1]))
    onehot[np.arange(comp.shape[0]), comp] = 1
    return onehot


def sim_true_effects(annotations, weights, cov_mats):
    """[num_pops, num_snps] effects, SNP i ~ N(0, cov_mats[component of i])
    (reference sim.py:96-133): component first, then one standard normal per (SNP, cohort),
    coloured by the Cholesky factor of the component's covariance."""
    cov_mats = np.asarray(cov_mats)
    comp = _component_index(annotations, weights)
    latent = np.random.normal(loc=0, scale=1, size=(comp.shape[0], cov_mats.shape[-1]))
    factors = np.linalg.cholesky(cov_mats)
    return np.einsum('iqp,ip->qi', factors[comp], latent)


def sim_gwas(true_beta, std_errs, ld_mat):
    """beta-hat = se * R (beta / se) + se * R^(1/2) eps, eps ~ N(0, I) (reference
    sim.py:136-156).  `ld_mat` is a BlockDiagonalMatrix; both products run on the GPU."""
    mean = std_errs * ld_mat.dot(true_beta / std_errs)
    eps = np.random.normal(loc=0, scale=1, size=true_beta.shape[0])
    return mean + std_errs * ld_mat.matrix_power(0.5).dot(eps)


def _load_weights(path, num_annotations, num_components):
    stored = np.load(path)
    weights = stored['hyper_delta'] if hasattr(stored, 'files') else np.array(stored)
    if weights.shape[0] != num_annotations:
        raise ValueError('The shape of the weights does not match the number of annotations.')
    if weights.shape[1] != num_components:
        raise ValueError('The shape of the weights does not match the number of covariance '
                         'matrices.')
    if not np.allclose(weights.sum(axis=1), 1.):
        raise ValueError('weights do not sum to 1 within each annotation.')
    return weights


def main(args):
    """Reference sim.py:159-272."""
    np.random.seed(args.seed)
    sumstat_paths = args.sumstats.split(',')
    num_pops = len(sumstat_paths)
    names = [str(p) for p in range(num_pops)]
    if args.names is not None:
        if args.names.count(',') != args.sumstats.count(','):
            raise ValueError('If --names are provided, one must be provided per sumstat file.')
        names = args.names.split(',')
    n_scales = np.ones(num_pops)
    n_scales[:] = np.array([float(v) for v in args.gwas_n_scaling.split(',')])
    if not np.all(n_scales > 0):
        raise ValueError('--gwas-n-scaling must be all positive.')

    # the SNPs of the simulation: union of the sumstats files, first occurrence wins
    all_vars = pd.concat([load.load_variant_list(path) for path in sumstat_paths],
                         ignore_index=True).drop_duplicates(subset='ID', ignore_index=True)
    num_snps = all_vars.shape[0]

    annotations, unannotated = load.load_annotations(args.annotations, all_vars)
    num_annotations = annotations.shape[1]
    # SNPs without an annotation get one at random, in proportion to the annotated ones
    share = annotations.sum(axis=0).astype(np.float64)
    share /= share.sum()
    drawn = np.random.choice(num_annotations, size=len(unannotated), p=share, replace=True)
    annotations[unannotated, :] = 0
    annotations[unannotated, drawn] = 1
    assert np.all(annotations.sum(axis=1) == 1)

    # only the standard errors of the sumstats are used; SNPs without data keep SE = 1e-100
    std_errs = np.full((num_pops, num_snps), 1e-100)
    ld_mats = []
    for p, (path, scale, schema) in enumerate(zip(sumstat_paths, n_scales,
                                                   args.ld_schema.split(','))):
        logging.info('Loading sumstats for population %s...', names[p])
        stats, no_stats = load.load_sumstats(path, all_vars)
        logging.info('Loading LD for population %s...', names[p])
        ld_mat, no_ld = load.load_ld_from_schema(schema, variants=all_vars, denylist=no_stats,
                                                 ldthresh=0.999999)
        ld_mats.append(ld_mat)
        keep = np.ones(num_snps, dtype=bool)
        keep[no_stats] = False
        keep[no_ld] = False
        std_errs[p, keep] = np.sqrt(1 / scale) * stats.SE.loc[keep]

    with open(args.covariance, 'rb') as handle:
        cov_mats = np.array(pickle.load(handle)[0])
    weights = _load_weights(args.weights, num_annotations, len(cov_mats))

    true_effects = sim_true_effects(annotations, weights, cov_mats)
    beta_hat = np.zeros((num_pops, num_snps))
    for p in range(num_pops):
        beta_hat[p] = sim_gwas(true_effects[p], std_errs[p], ld_mats[p])

    for p in range(num_pops):
        logging.info('Saving results for cohort %s', names[p])
        table = all_vars.copy()
        table['SE'] = std_errs[p]
        table['BETA'] = beta_hat[p]
        table['true_beta'] = true_effects[p]
        table.loc[table.SE < 1e-99, 'SE'] = np.nan
        table.dropna().to_csv('%s.%s.simgwas.tsv' % (args.output, names[p]), sep='\t',
                              index=False)
