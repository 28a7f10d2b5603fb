"""`vilma fit`: command-line surface, mixture-grid construction and output writers.

Flags, defaults, RNG consumption order and output files (.covariance.pkl, .npz,
.estimates.tsv, checkpoints) follow /root/reference/src/vilma/vi_options.py:9-337 so the
command is a drop-in; the optimisation itself runs on MI355X (variational_inference.py here).
Extra flag: --ld-form {auto,dense,eig} chooses how LD blocks are held in HBM.
"""
import itertools
import logging
import pickle

import numpy as np

from . import load, npz_writer


def args(super_parser):
    parser = super_parser.add_parser(
        'fit',
        description='Use variational inference to learn effect sizes and effect size '
                    'distribution from GWAS summary data (MI355X build).',
        usage='vilma fit <options>',
    )
    add = parser.add_argument
    add('-K', '--components', default=12, type=int,
        help='number of mixture components in prior')
    add('--num-its', default=1000, type=int, help='Maximum number of optimization iterations.')
    add('--ld-schema', required=True, type=str, help='Comma-separated paths to LD panel schemas.')
    add('--sumstats', required=True, type=str, help='Comma-separated paths to summary statistics.')
    add('--stderrscale', default='1.0', type=str, required=False,
        help='Comma separated list of values to multiply summary stat stderrs by.')
    add('--annotations', type=str, default=None, help='Path to annotation file.')
    add('--output', required=True, type=str, help='Output path prefix.')
    add('--names', type=str, required=False,
        help='Comma-separated names of the populations for output. Defaults to 0, 1,... ')
    add('--extract', required=True, type=str,
        help='List of SNPs to include in analysis, with ID, A1, and A2 columns.')
    add('--scaled', dest='scaled', action='store_true',
        help='Place the prior on frequency-scaled effect sizes instead of natural-scale ones.')
    add('--ldthresh', required=False, default=1.0, type=float,
        help='Threshold for singular value approximation of LD matrix: SNPs with an r^2 of x '
             'or larger stay linearly independent; 1 means no thresholding.')
    add('--seed', type=int, default=42, help='Seed for random number generation.')
    add('--mmap', dest='mmap', action='store_true',
        help='(unsupported on MI355X: LD is resident in HBM) store the LD matrix on disk.')
    add('--learn-scaling', dest='scale_se', action='store_true',
        help='Whether or not to learn a scaling factor for the standard errors.')
    add('--samplesizes', type=str, default='100e3',
        help='Comma-separated GWAS sample sizes used when initializing.')
    add('--init-hg', type=str, default='0.1',
        help='Comma-separated heritabilities per population, used only for initializing.')
    add('--trait', dest='trait', action='store_true',
        help='Treat sumstats files as different traits. Currently unimplemented.')
    add('--checkpoint-freq', type=int, default=-1,
        help='Store the model once every this many iterations. Defaults to no checkpointing.')
    add('--load-checkpoint', type=str, default='', nargs=2,
        metavar=('CHECKPOINT_FILE.npz', 'COVARIANCE_FILE.pkl'),
        help='Resume from a saved checkpoint (.npz) and its covariance matrices (.pkl).')
    add('--ld-form', type=str, default='auto', choices=['auto', 'dense', 'eig'],
        help='How LD blocks are held in HBM: dense reconstruction, eigen form, or per-block '
             'choice by bytes (default).')
    return parser


def _split_floats(text, n):
    out = np.zeros(n)
    out[:] = list(map(float, text.split(',')))
    return out


def _effect_range(betas, std_errs, scaled):
    """Plausible smallest/largest squared effect per population (vi_options.py:196-226)."""
    P = betas.shape[0]
    mins, maxes = np.zeros(P), np.zeros(P)
    if scaled:
        maxes = np.nanmax((betas / std_errs) ** 2, axis=1)
        for p in range(P):
            nz = betas[p, :] ** 2 > 0
            mins[p] = np.nanpercentile((betas[p, nz] / std_errs[p, nz]) ** 2, 2.5)
        return mins, maxes
    for p in range(P):
        ok = ~np.isnan(betas[p])
        b, se = np.abs(betas[p, ok]), std_errs[p, ok]
        psi = 1. / len(b)
        probs = 1. / (1. + ((1. - psi) / psi * np.sqrt(b ** 2 / se ** 2)
                            * np.exp(-0.5 * b ** 2 / se ** 2 + 0.5)))
        ebayes = np.maximum(b ** 2 - se ** 2, 1e-10)
        raw = b / (1. + se ** 2 / ebayes ** 2)
        maxes[p] = np.max(probs * raw) ** 2
        mins[p] = np.nanpercentile(betas[p, betas[p, :] ** 2 > 0] ** 2, 2.5)
    return mins, maxes


def _make_diag_vals(num_pops, num_components, mins, maxes):
    """Geometric grid of variances per population, preceded by a near-zero level
    (vi_options.py:284-298)."""
    levels = [[m * 1e-6 for m in mins]]
    for k in range(num_components + 1):
        levels.append([mins[p] * np.exp(np.log(maxes[p] / mins[p]) / num_components * k)
                       for p in range(num_pops)])
    return levels


def _jitter(num_pops):
    """One draw of the random rescaling the grid applies (consumes num_pops uniforms)."""
    return np.diag(np.sqrt(np.exp(np.random.uniform(-1, 1, num_pops))))


def _make_simple(num_pops, num_components, mins, maxes):
    """The reference's grid of mixture covariances (vi_options.py:301-337), with the same
    order of np.random draws so a seed gives the same grid."""
    levels = _make_diag_vals(num_pops, num_components, mins, maxes)
    if num_pops == 1:
        return list(np.array(levels).reshape((num_components + 2, 1, 1)))
    corr_grid = [-.99 + 1.98 * (k + 1) / num_components for k in range(num_components)]
    n_pairs = (num_pops * (num_pops - 1)) // 2
    upper = np.triu_indices(num_pops, k=1)
    covs = []
    for level_idx, diag in enumerate(levels):
        root = np.sqrt(diag)
        for corrs in itertools.product(*[corr_grid] * n_pairs):
            corr = np.eye(num_pops)
            corr[upper] = corrs
            corr.T[upper] = corrs
            base = (corr * root).T * root
            for _ in range(3):
                scale = _jitter(num_pops)
                covs.append(scale.dot(base.dot(scale)))
        if level_idx > 0:
            for p in range(num_pops):           # population-specific causal variants
                single = np.copy(levels[0])
                single[p] = diag[p]
                base = np.diag(single)
                for _ in range(3):
                    scale = _jitter(num_pops)
                    covs.append(scale.dot(base.dot(scale)))
    return covs


def main(args):
    from .sharding import init_distributed_from_env
    init_distributed_from_env()          # torchrun --nproc-per-node G: one rank per GPU
    np.random.seed(args.seed)
    n_schema_commas = args.ld_schema.count(',')
    if (not args.trait and n_schema_commas != 1
            and n_schema_commas != args.sumstats.count(',')):
        raise ValueError('Either need to imput one ld_schema or provide a sumstats file for '
                         'each ld_schema.')
    if args.trait:
        raise NotImplementedError('--trait has not been implemented yet.')
    sumstat_paths = args.sumstats.split(',')
    num_pops = len(sumstat_paths)
    names = list(map(str, range(num_pops)))
    if args.names is not None:
        if args.names.count(',') != args.sumstats.count(','):
            raise ValueError('If --names are provided, one must be provided per sumstat file.')
        names = args.names.split(',')

    logging.info('Loading variants...')
    variants = load.load_variant_list(args.extract)
    logging.info('Loading annotations...')
    annotations, denylist = load.load_annotations(args.annotations, variants=variants)
    n_snps = len(annotations)
    missing_annot = np.zeros(n_snps, dtype=bool)
    missing_annot[denylist] = True
    missing_sumstats = np.zeros((n_snps, num_pops), dtype=bool)
    missing_ld_info = np.zeros((n_snps, num_pops), dtype=bool)

    stderr_mult = _split_floats(args.stderrscale, num_pops)
    gwas_n = _split_floats(args.samplesizes, num_pops)
    init_hg = _split_floats(args.init_hg, num_pops)

    ld_mats, beta_rows, se_rows = [], [], []
    for idx, (schema_path, stats_path) in enumerate(zip(args.ld_schema.split(','),
                                                        sumstat_paths)):
        logging.info('Loading sumstats for population %d...', idx + 1)
        sumstats, missing = load.load_sumstats(stats_path, variants=variants)
        missing_sumstats[missing, idx] = True
        missing.extend(denylist)
        beta_rows.append(np.array(sumstats.BETA).reshape((1, -1)))
        logging.info('Largest beta is... %f', np.max(np.abs(np.array(sumstats.BETA))))
        se_rows.append(np.array(sumstats.SE).reshape((1, -1)) * stderr_mult[idx])
        logging.info('Loading LD for population %d...', idx + 1)
        ld_mat, no_ld = load.load_ld_from_schema(schema_path, variants=variants,
                                                 denylist=missing, ldthresh=args.ldthresh,
                                                 mmap=args.mmap, lazy=True)
        ld_mats.append(ld_mat)
        missing_ld_info[no_ld, idx] = True
    logging.info('Largest beta is... %f', np.max(np.abs(beta_rows)))
    betas = np.concatenate(beta_rows, axis=0)
    std_errs = np.concatenate(se_rows, axis=0)

    rank = 0
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
    except ImportError:
        pass

    if args.load_checkpoint:
        with open(args.load_checkpoint[1], 'rb') as pfile:
            cross_pop_covs = pickle.load(pfile)[0]
    else:
        logging.info('Building cross-population covariances...')
        mins, maxes = _effect_range(betas, std_errs, args.scaled)
        cross_pop_covs = _make_simple(num_pops, args.components, mins, maxes)
        if rank == 0:
            with open('%s.covariance.pkl' % args.output, 'wb') as ofile:
                pickle.dump([cross_pop_covs], ofile)

    logging.info('Fitting...')
    from .variational_inference import MultiPopVI
    elbo = MultiPopVI(
        marginal_effects=betas, std_errs=std_errs, ld_mats=ld_mats,
        mixture_covs=cross_pop_covs, annotations=annotations,
        checkpoint=(args.checkpoint_freq > 0), checkpoint_freq=args.checkpoint_freq,
        output=args.output, scaled=args.scaled, scale_se=args.scale_se, gwas_N=gwas_n,
        init_hg=init_hg, num_its=args.num_its, form=getattr(args, 'ld_form', 'auto'),
    )
    checkpoint = np.load(args.load_checkpoint[0]) if args.load_checkpoint else None
    params = elbo.optimize(checkpoint)

    to_save = elbo.create_dump_dict(params)
    to_save['vi_sigma'] = elbo.vi_sigma
    post_mean = elbo.real_posterior_mean(params)
    post_var = elbo.real_posterior_variance(params)
    if rank != 0:
        return
    # (numpy.savez's file, written by a thread pool: 34 GB with the default grid at 1 M SNPs)
    npz_writer.savez(args.output, **to_save)
    for name, row in zip(names, post_mean):
        variants['posterior_' + name] = row
    for name, row in zip(names, post_var):
        variants['posterior_variance_' + name] = row
    if args.annotations:
        variants['missing_annotation'] = missing_annot
    for idx, name in enumerate(names):
        variants['missing_sumstats_' + name] = missing_sumstats[:, idx]
        variants['missing_LD_' + name] = missing_ld_info[:, idx]
    variants.to_csv(args.output + '.estimates.tsv', sep='\t', index=False)
