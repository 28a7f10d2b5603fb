"""MultiPopVI for MI355X: the reference's class API over hand-written HIP kernels.

Interface (constructor keywords, `optimize`, `elbo`, `real_posterior_mean/variance`,
`create_dump_dict`, attributes, error behaviour) follows
/root/reference/src/vilma/variational_inference.py:96-109, 340-394, 599-630.  What differs is
where the work happens:

  * one outer iteration -- the line search on L, every accept/reject decision, the M-step, the
    error-scaling update (variational_inference.py:396-450, 762-802, 825-860) -- is ONE call into
    the library (vilma_sweep, include/vilma_hip.h), reproduced there decision for decision; this
    module keeps what surrounds it: argument checks, the optimize() loop with its checkpoints,
    convergence test and log lines, parameter tuples;
  * every objective evaluation is ONE fused per-SNP kernel + ONE block-diagonal LD product per
    cohort + a fixed-order reduction on the GPU, giving the 3P+2 sums of include/vilma_hip.h
    from which the objective is assembled;
  * state lives in HBM: vi_mu only.  vi_delta is a pure function of (vi_mu, hyper_delta,
    error_scaling) (_nat_to_not_vi_delta, variational_inference.py:632-641) and is
    re-derived inside the kernels; vi_sigma & co. are never materialised on the device;
  * R z of the accepted point is cached, so a sweep costs one LD product per *distinct*
    candidate point (the reference recomputes 5-8 per sweep);
  * with torch.distributed initialised, SNPs are sharded over ranks (one GPU each) and the
    sums are all-reduced before every decision -- by an RCCL communicator the context owns --
    so all ranks take the same branch.
"""
import logging
import os
import math

import numpy as np

from . import matrix_structures, npz_writer
from .sharding import Comm, plan_shards, local_ld

REL_TOL = 1e-6      # reference variational_inference.py:18-24 (the line search's own constants
ABS_TOL = 1e-6      # live in csrc/sweep.hip)
ELBO_TOL = 0.1
ELBO_MOMENTUM = 0.5
EPSILON = 1e-100    # reference numerics.py:8


def _inv_small(mats):
    """Inverse of [..., P, P] matrices; closed forms for P<=2 as the reference's helpers
    (numerics.py:216-244)."""
    P = mats.shape[-1]
    if P == 1:
        return 1.0 / mats
    if P == 2:
        a, b, c, d = mats[..., 0, 0], mats[..., 0, 1], mats[..., 1, 0], mats[..., 1, 1]
        r = 1.0 / (a * d - b * c)
        out = np.empty_like(mats)
        out[..., 0, 0] = d * r
        out[..., 1, 1] = a * r
        out[..., 1, 0] = -c * r
        out[..., 0, 1] = out[..., 1, 0]
        return out
    return np.linalg.inv(mats)


def _logdet_small(mats):
    P = mats.shape[-1]
    if P == 1:
        return np.log(mats[..., 0, 0])
    if P == 2:
        return np.log(mats[..., 0, 0] * mats[..., 1, 1] - mats[..., 0, 1] * mats[..., 1, 0])
    return np.linalg.slogdet(mats)[1]


def initial_hyper(delta_sums):
    """hyper_delta of _initialize (variational_inference.py:667-674) from the global sums."""
    hyper = np.asarray(delta_sums, dtype=np.float64) + 1.
    hyper /= hyper.sum(axis=1, keepdims=True)
    return np.maximum(hyper, EPSILON)


class DeviceParams:
    """(vi_mu, vi_delta, hyper_delta) of a state held on the GPU; arrays are downloaded (and
    gathered across ranks) only when indexed, so sweeps do not pay PCIe traffic."""

    def __init__(self, owner, version):
        self._owner, self._version = owner, version
        self._cache = {}

    @property
    def hyper_delta(self):
        """[A, M]; host-side state, fetched from the library the first time it is looked at."""
        if 'hyper' not in self._cache:
            if self._owner._version != self._version:
                raise RuntimeError('these parameters are no longer resident on the device; '
                                   'index the tuple before taking further steps')
            self._cache['hyper'] = np.array(self._owner._hyper)
        return self._cache['hyper']

    def _fetch(self, which):
        if which not in self._cache:
            if self._owner._version != self._version:
                raise RuntimeError('these parameters are no longer resident on the device; '
                                   'index the tuple before taking further steps')
            self._cache[which] = self._owner._download(which)
        return self._cache[which]

    def __len__(self):
        return 3

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(3))[i]
        if i < 0:
            i += 3
        if i == 0:
            return self._fetch('vi_mu')
        if i == 1:
            return self._fetch('vi_delta')
        if i == 2:
            return self.hyper_delta
        raise IndexError(i)

    def __iter__(self):
        return iter((self[0], self[1], self[2]))


class SweepDriver:
    """The host side of the fit loop over an engine holding one shard on one GPU.

    The sweep itself -- line search on L, accept / reject, M-step, error-scaling update, running
    ELBO change (variational_inference.py:396-450, 762-802, 825-860) -- runs inside the library
    (vilma_sweep, include/vilma_hip.h); what stays here is what the reference's class API exposes
    around it: parameter tuples, the optimize() loop with its checkpoints, convergence test and
    log lines.  Needs only the shard-independent constants (chi_stat, ld_ranks,
    annotation_counts, log_det) plus an engine and a Comm; MultiPopVI builds these from the
    reference's class-API inputs, bench.py builds them from device-resident synthetic data."""

    param_names = ['vi_mu', 'vi_delta', 'hyper_delta']

    def _setup_driver(self, engine, comm, num_pops, num_mix, num_annotations, chi_stat,
                      ld_ranks, annotation_counts, log_det, scale_se, num_its,
                      checkpoint=False, checkpoint_freq=-1, checkpoint_path='vilma-checkpoint'):
        self.engine, self.comm = engine, comm
        self.num_pops, self.num_mix, self.num_annotations = num_pops, num_mix, num_annotations
        self.chi_stat = np.asarray(chi_stat, dtype=np.float64)
        self.ld_ranks = np.asarray(ld_ranks, dtype=np.float64)
        self.annotation_counts = np.asarray(annotation_counts, dtype=np.float64)
        self.log_det = np.asarray(log_det, dtype=np.float64)
        self.scale_se, self.num_its = scale_se, num_its
        self.checkpoint, self.checkpoint_freq = checkpoint, checkpoint_freq
        self.checkpoint_path = checkpoint_path
        if not hasattr(self, 'error_scaling'):
            self.error_scaling = np.ones(num_pops)
        self._nat_table = None
        self._given = None          # evaluation at a caller-supplied vi_delta (see _upload)
        self._last_diff = None      # convergence statistics of the sweep just finished
        self._verbose = False
        self._look_ok = False       # the caller promises another sweep after this one
        self._veto = self._veto_next = False    # convergence vetoes of this / the next sweep
        self._version = 0           # bumped whenever the device state moves
        self._hyper_stale = False
        self._hyper_value = None
        self._objective = None
        self.n_evaluations = 0      # candidate points whose objective was looked at
        self.n_products = 0         # passes over the LD store (a two-step trial is one)
        self.n_trials = 0           # beta line-search trials among them
        self.n_stages_ahead = 0     # sweeps decided on the device, ahead of the host
        self.n_stages_skipped = 0   # stages queued ahead whose decision went the other way
        self.num_its_run = 0
        self.engine.set_annotation_counts(self.annotation_counts)
        self.engine.set_fit_constants(self.chi_stat, self.ld_ranks, scale_se)
        self.engine.bind_comm(comm)

    def start_from(self, vi_mu_local, hyper):
        """Make (vi_mu of this shard, hyper_delta) the current state and evaluate it.
        vi_mu_local = None: the device already holds vi_mu."""
        self._given = None
        self._objective = self.engine.set_state(vi_mu_local, hyper, self.error_scaling)
        self._install_hyper(hyper)
        self._version += 1
        return self._params()

    def initialize_from(self, fake_mu_local):
        """_initialize from the jittered ridge start of this shard's SNPs [P, N_local]: the
        per-SNP part (heuristic responsibilities, vi_mu; variational_inference.py:658-692), the
        all-reduce of the responsibility sums, hyper_delta (:667-674) and the first evaluation
        run behind vilma_initialize -- no [M,P,N] array on the host."""
        self._given = None
        self.engine.set_tau(self.error_scaling)
        self._objective = self.engine.initialize(fake_mu_local)
        self._install_hyper(self.engine.get_hyper())
        self._version += 1
        return self._params()

    # ------------------------------------------------------------------ device plumbing
    def _install_hyper(self, hyper):
        """Host-side copy of the hyper_delta installed on the device."""
        self._hyper = np.array(hyper, dtype=np.float64).reshape(self.num_annotations, self.num_mix)
        self._nat_table = None          # derived on demand (nat_grad_vi_delta)

    def _nat_table_of_hyper(self):
        """fast_vi_delta_grad (numerics.py:149-164) as an [A, M-1] table; the per-SNP array the
        reference stores is this table indexed by annotation.  Off the sweep's path: the device
        keeps its own copy (vilma_set_hyper / the M-step)."""
        if self._nat_table is None and self._hyper is not None:
            log_h = np.log(self._hyper) - 0.5 * self.log_det[None, :]
            self._nat_table = log_h[:, :-1] - log_h[:, -1:]
        return self._nat_table

    def _upload(self, params):
        """Make `params` the current device state and evaluate it.

        The device keeps vi_delta implicit -- it is the fixed point _nat_to_not_vi_delta(vi_mu,
        hyper_delta, error_scaling) (variational_inference.py:632-641), which is what every state
        the reference's own loop produces carries.  A vi_delta handed in from outside is checked
        against it: if it differs, the point (vi_mu, that vi_delta, hyper_delta) is evaluated as
        given and kept in self._given for elbo() / real_posterior_*(), which the reference defines
        on whatever they are handed (:412-417, 740-751)."""
        if isinstance(params, DeviceParams) and params._owner is self \
                and params._version == self._version:
            self._given = None
            return
        vi_mu, hyper = params[0], params[2]
        self.start_from(self._local_part(np.asarray(vi_mu)), hyper)
        # a DeviceParams' vi_delta was derived by the device itself: nothing to check (and
        # indexing it would download [N, M])
        delta = None if isinstance(params, DeviceParams) else params[1]
        if delta is not None:
            self._check_given_delta(np.asarray(delta, dtype=np.float64))

    GIVEN_DELTA_ATOL = 1e-9     # on entries of vi_delta, which lie in [1e-100, 1]

    def _objective_from(self, t):
        """fast_likelihood (numerics.py:31-46) minus _beta_KL (variational_inference.py:873-885)
        from all-reduced sums (only for points evaluated at a caller-supplied vi_delta; the
        sweep's own objectives are assembled inside the library)."""
        P = self.num_pops
        t = t.tolist()
        tau, chi, rk = self.error_scaling, self.chi_stat, self.ld_ranks
        lik = 0.0
        for p in range(P):
            lik += ((-0.5 * (t[P + p] + t[2 * P + p]) + t[p] - 0.5 * chi[p]) / tau[p]
                    - 0.5 * rk[p] * math.log(tau[p]))
        return lik - (t[3 * P] + t[3 * P + 1])

    def _check_given_delta(self, vi_delta):
        P = self.num_pops
        if vi_delta.shape != (self._num_snps_global(), self.num_mix):
            raise ValueError('vi_delta has the wrong shape.')
        out = self.engine.eval_given_delta(self._local_rows(vi_delta))
        nt = 3 * P + 2
        if self.comm.active:
            self.comm.allreduce_inplace(out[:nt])
            self.comm.allreduce_inplace(out[nt:], op='max')
        host = out.cpu().numpy()
        if not host[nt] > self.GIVEN_DELTA_ATOL:        # consistent (NaN counts as inconsistent)
            return
        mean, var = self.engine.get_trial_moments()
        self._given = {'objective': self._objective_from(host[:nt]), 'mean': mean, 'var': var,
                       'max_dev': float(host[nt])}

    def _require_fixed_point(self, what):
        """Entry points that CONTINUE from a state (sweeps, checkpoints) can only start from the
        fixed-point vi_delta; say so when the one handed in was something else."""
        if getattr(self, '_given', None) is not None:
            dev = self._given['max_dev']
            self._given = None
            logging.warning('%s: the vi_delta provided differs from the coordinate-ascent fixed '
                            'point of (vi_mu, hyper_delta, error_scaling) by up to %.3e; the '
                            'device state keeps vi_delta implicit, so the fit continues from the '
                            'fixed point (the reference would replace it at its first update).',
                            what, dev)

    def _local_part(self, vi_mu_global):
        return vi_mu_global

    def _local_rows(self, vi_delta_global):
        return vi_delta_global

    def _num_snps_global(self):
        return self.engine.N

    def _params(self):
        return DeviceParams(self, self._version)

    def _download(self, which):
        """This shard's vi_mu / vi_delta (MultiPopVI gathers them across ranks)."""
        self.engine.drain()         # nothing may be running ahead of the state being read
        if which == 'vi_mu':
            return self.engine.get_mu()
        return self.engine.get_delta()

    # ------------------------------------------------------------------ one sweep
    def _sweep_flags(self, want_diff):
        from . import _lib
        flags = _lib.SWEEP_DIFF if want_diff else 0
        if self._verbose:
            flags |= _lib.SWEEP_VERBOSE
        elif self._look_ok and os.environ.get('VILMA_LOOKAHEAD', '1') != '0':
            flags |= _lib.SWEEP_LOOKAHEAD
            if self._veto:
                flags |= _lib.SWEEP_VETO
            if self._veto_next:
                flags |= _lib.SWEEP_VETO_NEXT
        return flags

    def _log_events(self, stats):
        """The reference's per-update INFO lines (variational_inference.py:424-449, 782, 484)."""
        for e in stats.events[:stats.n_events]:
            if e.kind == 0:
                logging.info('...Updating paramset %d, L=%f', e.paramset, e.a)
            elif e.kind == 1:
                logging.info('...Old objective = %f, new objective = %f', e.a, e.b)
            elif e.kind == 2:
                logging.info('...Updating error_scaling, old ELBo=%f, new ELBo=%f', e.a, e.b)

    def _optimize_step(self, params, L, curr_elbo, line_search_rate=1.25,
                       running_elbo_delta=None, want_diff=False):
        """variational_inference.py:396-410 -- one call of vilma_sweep."""
        if hasattr(self.engine, 'refresh_stream'):
            self.engine.refresh_stream()
        self._upload(params)
        self._require_fixed_point('_optimize_step')
        L, elbo, running = self._step(L, curr_elbo, running_elbo_delta, line_search_rate, want_diff)
        return self._params(), L, elbo, running

    def _step(self, L, curr_elbo, running, line_search_rate, want_diff):
        """One vilma_sweep from the state the device holds; the host-side copies of what it changed
        (hyper_delta on demand)."""
        if self._hyper_value is None and not self._hyper_stale:
            raise RuntimeError('nat_grad_vi_delta must always be set prior to running '
                               '_update_beta')
        log_info = logging.getLogger().isEnabledFor(logging.INFO)
        if log_info:
            logging.info('Current ELBO = %f and L = %f,%f,%f,%f,%f', curr_elbo, *L[:5])
        if L.dtype != np.float64 or not L.flags.c_contiguous:
            L = np.ascontiguousarray(L, dtype=np.float64)
        was_verbose, self._verbose = self._verbose, self._verbose or log_info
        try:
            elbo, running, stats = self.engine.sweep(L, curr_elbo, running, line_search_rate,
                                                     self._sweep_flags(want_diff))
        except RuntimeError as exc:
            # the reference's line-search failure (variational_inference.py:790-799)
            if 'Encountered a numerical error.' in str(exc):
                raise RuntimeError('Encountered a numerical error.') from exc
            raise
        finally:
            self._verbose = was_verbose
        if log_info:
            self._log_events(stats)
        self.n_evaluations += stats.n_evaluations
        self.n_trials += stats.n_trials
        self.n_products += stats.n_products
        self.n_stages_ahead += stats.ran_ahead
        self.n_stages_skipped += stats.skipped_ahead
        if self.scale_se:
            self.error_scaling = np.array(stats.error_scaling[:self.num_pops])
        self._last_diff = (np.array(stats.diff_sum[:] + stats.diff_max[:]) if want_diff else None)
        self._hyper_stale = True        # fetched from the library when somebody looks at it
        self._nat_table = None
        self._objective = stats.objective
        self._version += 1
        return L, elbo, running

    @property
    def _hyper(self):
        if self._hyper_stale:
            self._hyper_stale = False
            self._hyper_value = self.engine.get_hyper()
        return self._hyper_value

    @_hyper.setter
    def _hyper(self, value):
        self._hyper_stale = False
        self._hyper_value = value

    def sweep(self, state=None, lookahead=False):
        """One outer iteration as optimize() runs it: _optimize_step + convergence statistics.
        `state` carries (L, elbo, running_elbo_delta) between calls; returns (state, stats).
        lookahead=True promises that sweep() is called again: the library may then queue the
        next sweep behind this one and decide its line search on the device (the state reported
        by this call is still the state after THIS sweep; the device may be one sweep further)."""
        if state is None:
            self.engine.snapshot_mean()
            state = {'L': np.ones(5), 'elbo': self._objective, 'running': None}
        self._look_ok, self._veto, self._veto_next = bool(lookahead), False, False
        self._given = None
        L, elbo, running = self._step(state['L'], state['elbo'], state['running'], 2., True)
        return {'L': L, 'elbo': elbo, 'running': running}, self._last_diff

    # ------------------------------------------------------------------ driver
    def optimize(self, loaded_checkpoint=None):
        """Initialise and run sweeps until convergence (variational_inference.py:340-394)."""
        if loaded_checkpoint is None:
            params = self._initialize()
        else:
            loaded = [loaded_checkpoint[name] for name in self.param_names]
            try:
                self.error_scaling = np.array(loaded_checkpoint['error_scaling'],
                                              dtype=np.float64)
            except KeyError:
                logging.warning('Did not find "error_scaling" in the loaded checkpoint. That '
                                'is okay, but we will have to assume that the error scalings '
                                'are 1.')
            self._upload(loaded)
            self._require_fixed_point('optimize(loaded_checkpoint)')
            params = self._params()
        converged = False
        elbo = self._objective
        running = None
        num_its = 0
        L = np.ones(5)
        n_total = self.num_pops * self.num_loci
        self.engine.snapshot_mean()
        verbose = logging.getLogger().isEnabledFor(logging.INFO)
        self._verbose, self._last_diff = verbose, None
        ckp_mean = self.real_posterior_mean(params) if (verbose and self.checkpoint) else None
        while num_its < self.num_its and not converged:
            if self.checkpoint and num_its % self.checkpoint_freq == 0:
                fname = '{}.{}'.format(self.checkpoint_path, num_its)
                dump = self.create_dump_dict(params)
                if self.comm.rank == 0:
                    npz_writer.savez(fname, **dump)
                if verbose:
                    ckp_mean = self.real_posterior_mean(params)
            # May the device run ahead of the host through the NEXT sweep?  Only if this sweep
            # cannot end the loop on a criterion the device does not see: the sweep count and the
            # running-ELBO rule (running_s >= MOMENTUM * running_{s-1}, so it stays above the
            # tolerance whenever MOMENTUM * |running| does); "no posterior mean moved" is vetoed
            # on the device itself; checkpoints need the state of their sweep.
            fresh_start = num_its < 10 and loaded_checkpoint is None
            self._veto = not fresh_start
            self._veto_next = not (num_its + 1 < 10 and loaded_checkpoint is None)
            self._look_ok = (num_its + 1 < self.num_its
                             and (fresh_start or (running is not None and
                                                  ELBO_MOMENTUM * abs(running) > ELBO_TOL))
                             and not (self.checkpoint and (num_its + 1) % self.checkpoint_freq == 0))
            params, L, elbo, running = self._optimize_step(
                params, L=L, curr_elbo=elbo, line_search_rate=2., running_elbo_delta=running,
                want_diff=True)
            d = self._last_diff
            converged = d[0] == 0
            converged = converged or abs(running) <= ELBO_TOL
            if num_its < 10 and loaded_checkpoint is None:
                converged = False
            if verbose:
                self._dump_info(num_its, d, n_total, params, ckp_mean)
            num_its += 1
        self._look_ok = False
        self.engine.drain()             # a sweep vetoed by convergence: forget it
        if num_its == self.num_its:
            logging.warning('Failed to converge')
        logging.info('Optimization ran for %d iterations', num_its)
        self.num_its_run = num_its
        return params

    def _dump_info(self, num_its, d, n_total, params, ckp_mean):
        """The reference's per-iteration INFO lines (variational_inference.py:292-331)."""
        logging.info('Completed iteration %d', num_its + 1)
        logging.info('Maximum posterior mean beta: %e', d[3])
        logging.info('SE scaling is: %r', self.error_scaling)
        logging.info('Max relative difference is: %e', d[5])
        logging.info('Max absolute difference is: %e', d[4])
        logging.info('Mean absolute difference is: %e', d[1] / n_total)
        logging.info('RMSE difference is: %e', np.sqrt(d[2] / n_total))
        if ckp_mean is not None:
            new = self.real_posterior_mean(params)
            logging.info('Max relative difference (checkpoint iterations) is: %e',
                         np.max(np.abs((new - ckp_mean) / (ckp_mean + EPSILON))))
            logging.info('Max absolute difference (checkpoint iterations) is: %e',
                         np.max(np.abs(new - ckp_mean)))
            logging.info('Mean absolute difference (checkpoint iterations) is: %e',
                         np.mean(np.abs(new - ckp_mean)))
            logging.info('RMSE difference (checkpoint iterations) is: %e',
                         np.sqrt(np.mean((new - ckp_mean) ** 2)))


class MultiPopVI(SweepDriver):
    """Fit the multi-population mixture-of-Gaussians VI scheme on MI355X.

    Keyword arguments are the reference's (variational_inference.py:96-142).  `form` selects
    the LD storage ('auto' | 'dense' | 'eig').  The engine is always the HIP engine
    (vilma_amd.engine.HipEngine): there is no CPU fallback and no parameter to supply another."""

    def __init__(self, marginal_effects=None, std_errs=None, ld_mats=None, annotations=None,
                 mixture_covs=None, checkpoint=True, checkpoint_freq=5, scaled=False,
                 scale_se=False, output='vilma_output', gwas_N=None, init_hg=None,
                 num_its=None, form='auto', _comm=None):
        # ---- argument checks, in the reference's order and with its exceptions ----
        if marginal_effects is not None and mixture_covs is not None:
            P_ = np.asarray(marginal_effects).shape[0]
            for mc in mixture_covs:
                if np.asarray(mc).shape != (P_, P_):
                    raise ValueError('Mixture component has a covariance matrix of the '
                                     'wrong shape.')
            signs, _ = np.linalg.slogdet(mixture_covs)
            if not np.all(signs == 1):
                raise ValueError('Mixture component has a non-positive definite covariance '
                                 'matrix.')
        required = (('init_hg', init_hg), ('gwas_N', gwas_N),
                    ('marginal_effects', marginal_effects), ('std_errs', std_errs),
                    ('ld_mats', ld_mats), ('annotations', annotations),
                    ('mixture_covs', mixture_covs), ('num_its', num_its))
        for name, val in required:
            if val is None:
                raise ValueError('%s must be specified when calling VIScheme()' % name)
        marginal_effects = np.asarray(marginal_effects, dtype=np.float64)
        std_errs = np.asarray(std_errs, dtype=np.float64)
        annotations = np.asarray(annotations)
        if not np.all(np.isfinite(marginal_effects)):
            raise ValueError('Encountered an infinite or NaN value in the GWAS effect size '
                             'estimates')
        if not np.all(np.isfinite(std_errs)):
            raise ValueError('Encountered an infinity or NaN value in the GWAS standard errors')

        self.scaled, self.scale_se = scaled, scale_se
        self.num_pops, self.num_loci = marginal_effects.shape
        self.num_mix = len(mixture_covs)
        P, N, M = self.num_pops, self.num_loci, self.num_mix
        self.error_scaling = np.ones(P)
        self.checkpoint, self.checkpoint_freq = checkpoint, checkpoint_freq
        self.checkpoint_path = '%s-checkpoint' % output
        if len(ld_mats) != P:
            raise ValueError('Fewer LD matrices than populations.')
        for ld in ld_mats:
            if not isinstance(ld, matrix_structures.BlockDiagonalMatrix):
                raise ValueError('LD Matrices must be of type BlockDiagonalMatrix.')
        for ld in ld_mats:
            if ld.shape != (N, N):
                raise ValueError('LD matrix shape does not match GWAS marginal effect size '
                                 'shape.')
        if not np.allclose(annotations.sum(axis=1), 1):
            raise ValueError('Some SNPs are either missing annotations or have more than one '
                             'annotation.')
        if annotations.shape[0] != N:
            raise ValueError('annotations dimension does not match GWAS marginal effect size '
                             'shape.')
        self.num_annotations = annotations.shape[1]
        if scaled:                                   # variational_inference.py:205-214
            self.marginal_effects = marginal_effects / (std_errs + EPSILON)
            self.std_errs = np.ones_like(std_errs)
            self.scalings = std_errs + EPSILON
        else:
            self.marginal_effects = np.copy(marginal_effects)
            self.std_errs = np.copy(std_errs)
            self.scalings = np.ones_like(std_errs)
        self.ld_mats = ld_mats
        self.annotations = np.copy(np.where(annotations)[1])
        self.annotation_counts = annotations.sum(axis=0)
        self.init_hg = np.asarray(init_hg, dtype=np.float64)
        self.gwas_N = np.asarray(gwas_N, dtype=np.float64)
        self.num_its = num_its

        covs = np.array(mixture_covs, dtype=np.float64)            # :621-626
        prec = _inv_small(covs)
        self.mixture_prec = prec[:, :, :, None]
        self.log_det = _logdet_small(covs)

        # ---- shard plan and the device engine for this rank ----
        self.comm = _comm if _comm is not None else Comm()
        per_snp = 8.0 * (3 * M * P)
        self._plan = plan_shards(ld_mats, N, self.comm.world, per_snp_cost=per_snp)
        mine = self._plan[self.comm.rank]
        self._snps = mine['snps']
        n_loc = len(self._snps)
        empty = [r for r, part in enumerate(self._plan) if len(part['snps']) == 0]
        if empty:
            # every rank computes the same plan, so every rank raises here, before any collective
            raise ValueError('rank(s) %s would receive no SNPs: fewer independent LD components '
                             'than GPUs (%d)' % (empty, self.comm.world))
        from . import engine as _engine
        self.engine = _engine.HipEngine(P, n_loc, M, self.num_annotations)

        # ---- one-time constants (variational_inference.py:189-252) for this shard's blocks;
        # the per-SNP results are gathered across ranks ----
        loc = self._snps
        z_loc = self.marginal_effects[:, loc] / self.std_errs[:, loc]
        inv_se2 = self.comm.allreduce_np((self.std_errs[:, loc] ** -2).sum(axis=1)) \
            if self.comm.active else (self.std_errs ** -2).sum(axis=1)
        prior = 2 * self.gwas_N * self.init_hg / inv_se2
        local_lds = []
        for p, ld in enumerate(ld_mats):
            mats, perm, n_ld = local_ld(ld, loc, mine['blocks'][p], N)
            local_lds.append((matrix_structures.BlockDiagonalMatrix(mats, perm=perm,
                                                                     missing=perm[n_ld:]),
                              perm, n_ld))
        if hasattr(self.engine, 'ld_begin'):
            consts = self._load_on_device(local_lds, z_loc, prior, form)
        else:
            consts = self._load_on_host(local_lds, z_loc, prior, form)
        diag_loc, adj_loc, inverse_loc, chi_loc, rank_loc = consts
        self.ld_diags = self.comm.gather_snps(diag_loc, loc, N)
        self.scaled_ld_diags = self.std_errs ** -2 * self.ld_diags
        sums = self.comm.allreduce_np(np.concatenate([chi_loc, rank_loc]))
        self.chi_stat, self.ld_ranks = sums[:P], sums[P:]
        self.adj_marginal_effects = self.comm.gather_snps(adj_loc, loc, N)
        self.inverse_betas = self.comm.gather_snps(inverse_loc, loc, N)
        if not np.allclose(self.adj_marginal_effects[np.isclose(self.ld_diags, 0)], 0):
            raise ValueError('Some SNPs that are missing in the LD matrix are not being '
                             'treated as missing.')

        self.engine.set_snp_data(adj_loc, self.std_errs[:, loc], self.scaled_ld_diags[:, loc],
                                 self.scalings[:, loc], self.annotations[loc])
        self.engine.set_mixture(prec, self.log_det)
        self.engine.set_tau(self.error_scaling)

        self._setup_driver(self.engine, self.comm, P, M, self.num_annotations, self.chi_stat,
                           self.ld_ranks, self.annotation_counts, self.log_det, scale_se, num_its,
                           checkpoint=checkpoint, checkpoint_freq=checkpoint_freq,
                           checkpoint_path=self.checkpoint_path)

    def _load_on_device(self, local_lds, z_loc, prior, form):
        """Load-time constants with the GPU doing everything but `eigh` (ld_device.py): blocks
        stream host -> device as they are decomposed; diag, R^+ z, chi, R R^+ z per block on the
        device; the ridge start by conjugate gradients on the resident LD store."""
        from . import ld_device
        P = self.num_pops
        loc = self._snps
        n_loc = len(loc)
        se = self.std_errs[:, loc]
        diag_loc, rmle = np.zeros((P, n_loc)), np.zeros((P, n_loc))
        chi_loc, rank_loc = np.zeros(P), np.zeros(P)
        for p, (sub, perm, n_ld) in enumerate(local_lds):
            out = ld_device.stream_cohort(self.engine, p, sub, form, z_loc[p][perm[:n_ld]])
            diag_loc[p, perm[:n_ld]] = out['diag']
            rmle[p, perm[:n_ld]] = out['rmle']
            chi_loc[p], rank_loc[p] = out['chi'], out['rank']
        adj_loc = rmle / se                                   # (R R^+ z) / se  (:243)
        reg = se ** 2 / prior[:, None]
        try:
            ridge = ld_device.ridge_start(self.engine, rmle, reg, diag_loc)   # rhs = adj * se
        except ld_device.RidgeStalled as exc:
            # the reference's own per-block solve (matrix_structures.py:349-387) always returns a
            # starting point: fall back to it on the host (re-decomposing blocks that were dropped)
            # (blocks the streaming loader dropped from host memory are decomposed again, on demand)
            again = sum(1 for sub, _, _ in local_lds for m in sub.matrices
                        if getattr(m, 'is_deferred', lambda: False)())
            logging.warning('%s; falling back to the per-block ridge solve on the host '
                            '(%d block(s) decomposed again)', exc, again)
            ridge = np.zeros((P, n_loc))
            for p, (sub, perm, n_ld) in enumerate(local_lds):
                ridge[p] = sub.ridge_inverse_dot(rmle[p], reg[p])
        return diag_loc, adj_loc, ridge * se, chi_loc, rank_loc

    def _load_on_host(self, local_lds, z_loc, prior, form):
        """The same constants with the reference's own per-block formulas in host numpy
        (matrix_structures.py) -- for engines without a device loader (the test engine)."""
        P = self.num_pops
        loc = self._snps
        n_loc = len(loc)
        se = self.std_errs[:, loc]
        subs = []
        for p, (sub, perm, n_ld) in enumerate(local_lds):
            sub.materialize()
            self.engine.load_ld(p, sub.device_blocks(form), perm, n_ld)
            subs.append(sub)
        diag_loc = np.stack([sub.diag() for sub in subs])
        mle = np.zeros((P, n_loc))
        chi_loc, rank_loc = np.zeros(P), np.zeros(P)
        for p in range(P):
            mle[p] = subs[p].inverse.dot(z_loc[p])
            chi_loc[p] = z_loc[p].dot(mle[p])
            rank_loc[p] = subs[p].get_rank()
        adj_loc = self.engine.ld_matvec(mle) / se
        inverse_loc = np.zeros((P, n_loc))
        for p in range(P):
            ridge = subs[p].ridge_inverse_dot(adj_loc[p] * se[p], se[p] ** 2 / prior[p])
            inverse_loc[p] = ridge * se[p]
        return diag_loc, adj_loc, inverse_loc, chi_loc, rank_loc

    @property
    def nat_grad_vi_delta(self):
        """[N, M-1] natural parameter of the mixture weights (numerics.py:149-164), expanded
        from the [A, M-1] table the device uses; None until hyper_delta has been set."""
        if getattr(self, '_hyper', None) is None:
            return None
        return self._nat_table_of_hyper()[self.annotations]

    @nat_grad_vi_delta.setter
    def nat_grad_vi_delta(self, value):
        if value is None:
            self._nat_table = self._hyper = None

    def _local_part(self, vi_mu_global):
        return vi_mu_global[:, :, self._snps]

    def _local_rows(self, vi_delta_global):
        return vi_delta_global[self._snps]

    def _num_snps_global(self):
        return self.num_loci

    def _download(self, which):
        self.engine.drain()         # nothing may be running ahead of the state being read
        if which == 'vi_mu':
            return self.comm.gather_snps(self.engine.get_mu(), self._snps, self.num_loci)
        delta = self.engine.get_delta()
        if self.comm.world == 1 and len(self._snps) == self.num_loci and np.array_equal(
                self._snps, np.arange(self.num_loci)):
            return delta                # [N, M] as the device wrote it: no transposed copies
        return np.ascontiguousarray(
            self.comm.gather_snps(np.ascontiguousarray(delta.T), self._snps, self.num_loci).T)

    # ------------------------------------------------------------------ host-side views
    def _lam(self):
        lam = np.empty((self.num_mix, self.num_pops, self.num_pops, self.num_loci))
        lam[:] = self.mixture_prec                      # [M,P,P,1] broadcast over SNPs
        diag = self.scaled_ld_diags / self.error_scaling.reshape((-1, 1))
        for p in range(self.num_pops):
            lam[:, p, p, :] = diag[p] + self.mixture_prec[:, p, p, :]
        return lam

    @property
    def vi_sigma(self):
        """[M,P,P,N], computed on demand for outputs (variational_inference.py:712-724): on the
        device when this rank holds every SNP in order (vilma_get_vi_sigma: the same closed forms;
        7 s of numpy at 1 M SNPs x 582 components otherwise), else on the host from the gathered
        constants."""
        if (self.comm.world == 1 and hasattr(self.engine, 'get_vi_sigma')
                and len(self._snps) == self.num_loci
                and np.array_equal(self._snps, np.arange(self.num_loci))):
            return self.engine.get_vi_sigma(self.error_scaling)
        return self._vi_sigma_host()

    def _vi_sigma_host(self):
        lam = self._lam()
        if self.num_pops == 1:
            return 1.0 / lam
        if self.num_pops == 2:
            # the closed form of _inv_small written on the [M,P,P,N] layout itself (same
            # arithmetic; the transposed views cost seconds at 1 M SNPs)
            a, b, c, d = lam[:, 0, 0], lam[:, 0, 1], lam[:, 1, 0], lam[:, 1, 1]
            r = 1.0 / (a * d - b * c)
            out = np.empty_like(lam)
            out[:, 0, 0] = d * r
            out[:, 1, 1] = a * r
            out[:, 1, 0] = -c * r
            out[:, 0, 1] = out[:, 1, 0]
            return out
        return np.transpose(_inv_small(np.transpose(lam, (3, 0, 1, 2))), (1, 2, 3, 0))

    @property
    def nat_sigma(self):
        return -0.5 * self._lam()

    @property
    def vi_sigma_log_det(self):
        return -np.transpose(_logdet_small(np.transpose(self._lam(), (3, 0, 1, 2))))

    @property
    def vi_sigma_matches(self):
        return np.einsum('kpq,kqpi->ik', self.mixture_prec[:, :, :, 0], self.vi_sigma)

    @property
    def sigma_summary(self):
        return self.log_det - self.vi_sigma_log_det.T + self.vi_sigma_matches

    # ------------------------------------------------------------------ public API
    def elbo(self, params):
        """ELBO of `params` (variational_inference.py:412-417), at the vi_delta given."""
        self._upload(params)
        return self._objective if self._given is None else self._given['objective']

    def _moments(self, params):
        self._upload(params)
        if self._given is not None:
            mean, var = self._given['mean'], self._given['var']
        else:
            self.engine.drain()
            mean, var = self.engine.get_moments()
        return (self.comm.gather_snps(mean, self._snps, self.num_loci),
                self.comm.gather_snps(var, self._snps, self.num_loci))

    def real_posterior_mean(self, vi_mu=None, vi_delta=None, hyper_delta=None):
        params = vi_mu if isinstance(vi_mu, DeviceParams) else (vi_mu, vi_delta, hyper_delta)
        return self._moments(params)[0] * self.scalings

    def real_posterior_variance(self, vi_mu=None, vi_delta=None, hyper_delta=None):
        params = vi_mu if isinstance(vi_mu, DeviceParams) else (vi_mu, vi_delta, hyper_delta)
        return self._moments(params)[1] * self.scalings ** 2

    def create_dump_dict(self, params):
        dump = dict(zip(self.param_names, (params[0], params[1], params[2])))
        dump['error_scaling'] = self.error_scaling
        dump['scalings'] = self.scalings
        return dump

    # ------------------------------------------------------------------ initialisation
    def _initialize(self):
        """Starting point (variational_inference.py:643-700).  The random draw is the
        reference's single legacy-RNG call over the full [P,N] array, so seeds reproduce."""
        P, N, M = self.num_pops, self.num_loci, self.num_mix
        logging.info('Largest inverse_beta is %f', np.max(np.abs(self.inverse_betas)))
        missing = np.isclose(self.ld_diags, 0)
        fake_mu = np.random.normal(loc=np.copy(self.inverse_betas),
                                   scale=1e-3 * self.std_errs, size=(P, N))
        fake_mu[missing] = np.nan
        n_obs = (~missing).sum(axis=0)
        col_mean = np.where(n_obs > 0, np.nansum(fake_mu, axis=0) / np.maximum(n_obs, 1), np.nan)
        fill = np.tile(col_mean, [P, 1])
        fake_mu[missing] = fill[missing]
        fake_mu[np.isnan(fake_mu)] = 0.

        return self.initialize_from(fake_mu[:, self._snps])

    def _set_state(self, params):
        self._upload(params)
        self._require_fixed_point('_set_state')

    # ------------------------------------------------------------------ the reference's private steps
    # Same names and argument order as the reference's methods, so code (and tests) written against
    # them keep working; each is one call into the library on the state `params` describes.
    def _posterior_mean(self, vi_mu, vi_delta, hyper_delta=None):
        """variational_inference.py:753-755 (without scalings)."""
        return self._moments((vi_mu, vi_delta, hyper_delta if hyper_delta is not None
                              else self._hyper))[0]

    def _posterior_marginal_variance(self, post_mean, vi_mu, vi_delta, hyper_delta=None):
        """variational_inference.py:757-760 (without scalings)."""
        return self._moments((vi_mu, vi_delta, hyper_delta if hyper_delta is not None
                              else self._hyper))[1]

    def _nat_to_not_vi_delta(self, params):
        """variational_inference.py:632-641: params with vi_delta replaced by the fixed point of
        (vi_mu, hyper_delta, error_scaling)."""
        vi_mu, _, hyper = params[0], params[1], params[2]
        self.start_from(self._local_part(np.asarray(vi_mu)), hyper)
        p = self._params()
        return (p[0], p[1], p[2])

    def _continue_from(self, params, what):
        self._upload(params)
        self._require_fixed_point(what)

    def _update_beta(self, vi_mu, vi_delta, hyper_delta, orig_obj, L, idx, lsr):
        """variational_inference.py:762-802."""
        self._continue_from((vi_mu, vi_delta, hyper_delta), '_update_beta')
        if self._hyper is None:
            raise RuntimeError('nat_grad_vi_delta must always be set prior to running '
                               '_update_beta')
        try:
            L[idx], orig, new = self.engine.update_beta(L[idx], lsr)
        except RuntimeError as exc:
            if 'Encountered a numerical error.' in str(exc):
                raise RuntimeError('Encountered a numerical error.') from exc
            raise
        self._objective = new
        self._version += 1
        p = self._params()
        return (p[0], p[1], p[2]), L, (orig if orig_obj is None else orig_obj), new

    def _update_hyper_delta(self, vi_mu, vi_delta, hyper_delta, orig_obj, L, idx, lsr):
        """variational_inference.py:825-860."""
        self._continue_from((vi_mu, vi_delta, hyper_delta), '_update_hyper_delta')
        orig, new = self.engine.update_hyper_delta()
        self._objective = new
        self._hyper_stale = True
        self._nat_table = None
        self._version += 1
        p = self._params()
        return (p[0], p[1], p[2]), L, (orig if orig_obj is None else orig_obj), new

    def _update_error_scaling(self, params):
        """variational_inference.py:472-486 (the sigma-dependent constants follow tau inside the
        kernels, so _set_vi_sigma is implied)."""
        self._continue_from(params, '_update_error_scaling')
        _, new = self.engine.update_error_scaling()
        self.error_scaling = self.engine.get_tau()
        self._objective = new
        self._version += 1

    def _set_vi_sigma(self):
        """variational_inference.py:712-733: vi_sigma & co. are functions of error_scaling here
        (properties); make the device follow a tau assigned from outside."""
        self.engine.set_tau(self.error_scaling)
        self._version += 1

    def _nat_grad_step(self, params, L, line_search_rate, running_elbo_delta=None):
        """variational_inference.py:419-450: (params, L, change of the ELBO)."""
        L = np.ascontiguousarray(list(L) + [1.] * (5 - len(L)), dtype=np.float64)
        new_params, L, elbo, _ = self._optimize_step(params, L, 0.0, line_search_rate,
                                                     running_elbo_delta)
        return (new_params[0], new_params[1], new_params[2]), L, elbo

