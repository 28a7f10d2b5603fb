"""MultiPopVI for MI355X: the reference's class API over hand-written HIP kernels.

Interface (constructor keywords, `optimize`, `elbo`, `real_posterior_mean/variance`,
`create_dump_dict`, attributes, error behaviour) follows
/root/reference/src/vilma/variational_inference.py:96-109, 340-394, 599-630.  What differs is
where the work happens:

  * the outer loop, the line search on L and every accept/reject decision stay on the host
    (variational_inference.py:340-450, 762-802) and are reproduced decision for decision;
  * every objective evaluation is ONE fused per-SNP kernel + ONE block-diagonal LD product per
    cohort + a fixed-order reduction on the GPU (libvilma_hip.so), returning the 3P+2 sums of
    include/vilma_hip.h from which the host assembles the objective;
  * state lives in HBM: vi_mu only.  vi_delta is a pure function of (vi_mu, hyper_delta,
    error_scaling) (_nat_to_not_vi_delta, variational_inference.py:632-641) and is
    re-derived inside the kernels; vi_sigma & co. are never materialised on the device;
  * R z of the accepted point is cached, so a sweep costs one LD product per *distinct*
    candidate point (the reference recomputes 5-8 per sweep);
  * with torch.distributed initialised, SNPs are sharded over ranks (one GPU each) and the
    sums are all-reduced (RCCL) before every decision, so all ranks take the same branch.
"""
import logging
import os
import math

import numpy as np

from . import matrix_structures
from .sharding import Comm, plan_shards, local_ld

L_MAX = 1e12        # reference variational_inference.py:18-24
REL_TOL = 1e-6
ABS_TOL = 1e-6
ELBO_TOL = 0.1
EM_TOL = 10
ELBO_MOMENTUM = 0.5
MAX_NUM_ITERS = 20
EPSILON = 1e-100    # reference numerics.py:8
# run ahead of a line-search decision only if its L exceeds the last rejected L by this factor
LOOKAHEAD_MARGIN = float(os.environ.get('VILMA_LOOKAHEAD_MARGIN', '1.1'))

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

    def __init__(self, owner, version, hyper):
        self._owner, self._version = owner, version
        self.hyper_delta = np.array(hyper)
        self._cache = {}

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
    """The host side of the sweep loop over an engine holding one shard on one GPU.

    Needs only the shard-independent constants (chi_stat, ld_ranks, annotation_counts, log_det)
    plus an engine and a Comm; MultiPopVI builds these from the reference's class-API inputs,
    bench.py builds them from device-resident synthetic data."""

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
        self._want_diff = False
        self.engine.set_annotation_counts(self.annotation_counts)
        self._last_diff = None      # convergence statistics fetched together with an evaluation
        self._cur_sums = None       # all-reduced responsibility sums of the current state, if known
        self._trial_sums = None
        self._pending = None        # a beta trial queued ahead of time: {'step', 'host'}
        self._speculate = False     # optimize()/sweep() loops: queue the next sweep's first trial
        self._verbose = False
        self._log_info = False
        self._views = {}            # slices of the engine's result vector handed to all-reduces
        # decisions on the device (engine.decide): the stage queued ahead of its decision
        self._look_ok = False       # the caller promises another standard sweep after this one
        self._veto = self._veto_next = False    # convergence vetoes of this / the next sweep
        self._ahead = None          # stage pre-queued for the NEXT sweep: {'pred','out'}
        self._mine = None           # this sweep's stage when it was pre-queued and confirmed
        self._pending_flag = None   # device decision that came with the pending trial
        self._half_rank_log_tau = None
        self._version = 0           # bumped whenever the device state moves
        self._hyper = None
        self._totals = None         # all-reduced sums of the current (accepted) state
        self._objective = None
        self.n_evaluations = 0      # candidate points whose objective was looked at
        self.n_products = 0         # passes over the LD store (a two-step trial is one)
        self._alt = None            # second candidate of the last two-step trial, not looked at yet
        self._candidate = 1
        self._lsr = 2.
        # Two-step beta trials (vilma_trial_beta2): measured +6...8 % sweeps/s at C3 and +2.5 % at
        # C2 (P = 2 / 1: the second candidate's per-SNP work is cheap next to the LD product it
        # saves on every rejected step), -2 % at C5 (P = 4: there the per-SNP pass is the larger
        # half of a trial) -- profiles/r02h_ab_twostep.txt.  VILMA_TWO_STEP=0/1 overrides.
        two = os.environ.get('VILMA_TWO_STEP')
        self._two_step = hasattr(self.engine, 'trial2') and (
            two == '1' if two in ('0', '1') else num_pops <= 2)
        self.n_trials = 0           # beta line-search trials among them
        self.n_stages_ahead = 0     # sweeps whose M-step stage ran ahead of the host's decision
        self.n_stages_skipped = 0   # stages queued ahead whose decision went the other way: their
                                    # kernels exited at once (a few microseconds each)
        self.num_its_run = 0

    def start_from(self, vi_mu_local, hyper):
        """Make (vi_mu of this shard, hyper_delta) the current state and evaluate it.
        vi_mu_local = None: the device already holds vi_mu (engine.init_state)."""
        self._drop_ahead()
        self._pending = None
        self._given = None
        self._half_rank_log_tau = None
        self.engine.set_tau(self.error_scaling)
        self._set_hyper(hyper)
        if vi_mu_local is not None:
            self.engine.set_mu(vi_mu_local)
        obj, totals = self._evaluate()
        self._accept(False, obj, totals)
        return self._params()

    def initialize_from(self, fake_mu_local):
        """_initialize from the jittered ridge start of this shard's SNPs [P, N_local]: the
        per-SNP part (heuristic responsibilities, vi_mu; variational_inference.py:658-692) runs
        on the device and leaves vi_mu there -- no [M,P,N] array on the host -- and only the
        [A,M] responsibility sums come back for hyper_delta (:667-674)."""
        self.engine.set_tau(self.error_scaling)
        sums = self.engine.init_state(fake_mu_local)
        hyper = initial_hyper(self.comm.allreduce(sums).reshape(self.num_annotations,
                                                                 self.num_mix))
        return self.start_from(None, hyper)

    # ------------------------------------------------------------------ device plumbing
    def _objective_from(self, t):
        """fast_likelihood (numerics.py:31-46) minus _beta_KL (variational_inference.py:873-885)
        from the all-reduced sums."""
        P = self.num_pops
        t = t.tolist()
        tau, chi, rk = self.error_scaling, self.chi_stat, self.ld_ranks
        lik = 0.0
        for p in range(P):          # P <= 4: plain floats beat small-array numpy here
            lik += ((-0.5 * (t[P + p] + t[2 * P + p]) + t[p] - 0.5 * chi[p]) / tau[p]
                    - 0.5 * rk[p] * math.log(tau[p]))
        return lik - (t[3 * P] + t[3 * P + 1])

    def _fetch(self, lo, hi, with_max=False):
        """All-reduce what has to be summed over ranks -- ONE RCCL all-reduce on the contiguous
        part [lo, hi) of the engine's result vector -- and download the whole vector in ONE
        device->host copy.  Returns the host copy; slices are in engine.layout."""
        eng = self.engine
        if self.comm.active:
            view = self._views.get((lo, hi))
            if view is None:                    # tensor slicing costs microseconds per call
                view = self._views[(lo, hi)] = eng.results[lo:hi]
            self.comm.allreduce_inplace(view)
            if with_max and self._verbose:
                self.comm.allreduce_inplace(eng.results[eng.layout.dmax], op='max')
        return eng.fetch()

    def _evaluate(self):
        """Objective of the CURRENT vi_mu under the current hyper/tau.  The evaluated point
        stays on the device as the trial state."""
        L = self.engine.layout
        self._drop_ahead()
        self._pending = None            # a queued trial's buffers are about to be overwritten
        self.engine.eval()
        host = self._fetch(L.totals.start, L.totals.stop)
        totals = host[L.totals]
        self._trial_sums = None
        self._alt = None
        self.n_evaluations += 1
        self.n_products += 1
        return self._objective_from(totals), totals

    def _launch_trial(self, step, alt_step=None, keep_sums=False):
        """Queue one natural-gradient trial at `step` plus the responsibility sums of the
        candidate (the M-step statistic), so an accepted candidate needs no second round trip.
        With `alt_step` -- the step the line search would try next if `step` is rejected -- both
        candidates are evaluated in the same pass over vi_mu and the LD store (the LD kernel is
        HBM-bound: the second right-hand side rides in the same loads), so a rejection no
        longer costs a second LD product.  This overwrites the device copy of the sums."""
        from . import _lib
        if not keep_sums:
            self._cur_sums = None
        if alt_step is not None and self._two_step:
            self.engine.trial2(step, alt_step)
        else:
            self.engine.trial(step)
        self.engine.delta_sums(_lib.STATE_TRIAL_BETA)

    def _trial(self, step, alt_step=None):
        """Objective of the candidate at `step`: the second candidate of the pair evaluated last
        if that is this step; else the result of the trial already queued behind the previous
        sweep's last evaluation if there is one for this step; else queue it now."""
        L = self.engine.layout
        alt, self._alt = self._alt, None
        if alt is not None and alt['step'] == step:
            self._candidate = 2
            self._trial_sums = None         # sums ride with candidate A only
            self._pending_flag = None
            self.n_evaluations += 1
            self.n_trials += 1
            return alt['obj'], alt['totals']
        pend, self._pending = self._pending, None
        self._pending_flag = None
        if pend is not None and pend['step'] == step:
            host = pend['host']
            self._pending_flag = pend.get('flag')
            alt_step = pend.get('alt_step')
        else:
            self._drop_ahead()
            if not self._two_step:
                alt_step = None
            self._launch_trial(step, alt_step)
            host = self._fetch(L.ttotals.start, L.sums.stop)
        totals = host[L.ttotals]
        self._trial_sums = host[L.sums]
        self._candidate = 1
        if alt_step is not None:
            tb = host[L.ttotals_b]
            self._alt = {'step': alt_step, 'obj': self._objective_from(tb), 'totals': tb}
        self.n_evaluations += 1
        self.n_trials += 1
        self.n_products += 1
        return self._objective_from(totals), totals

    def _accept(self, take_mu, obj, totals, already_flipped=False):
        self._alt = None                    # whatever second candidate there was is gone now
        if not already_flipped:
            self.engine.accept(take_mu)
        self._objective, self._totals = obj, totals
        # responsibility sums fetched with the candidate now describe the current state
        self._cur_sums, self._trial_sums = self._trial_sums, None
        self._version += 1

    def _install_hyper(self, hyper):
        """Host-side bookkeeping of a hyper_delta that is already installed on the device."""
        self._hyper = np.array(hyper)
        self._nat_table = None          # derived on demand (nat_grad_vi_delta)

    def _nat_table_of_hyper(self):
        """fast_vi_delta_grad (numerics.py:149-164) as an [A, M-1] table; the per-SNP array the
        reference stores is this table indexed by annotation.  Off the sweep's critical path: the
        device keeps its own copy (vilma_set_hyper / vilma_mstep)."""
        if self._nat_table is None and self._hyper is not None:
            log_h = np.log(self._hyper) - 0.5 * self.log_det[None, :]
            self._nat_table = log_h[:, :-1] - log_h[:, -1:]
        return self._nat_table

    def _set_hyper(self, hyper):
        self.engine.set_hyper(np.asarray(hyper, dtype=np.float64))
        self._install_hyper(hyper)

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
        return DeviceParams(self, self._version, self._hyper)

    def _download(self, which):
        """This shard's vi_mu / vi_delta (MultiPopVI gathers them across ranks)."""
        if which == 'vi_mu':
            return self.engine.get_mu()
        self._pending = None        # vilma_get_delta uses the trial vi_mu buffer as scratch
        return self.engine.get_delta()

    # ------------------------------------------------------------------ one sweep
    def _update_beta(self, L, idx, lsr, orig_obj):
        """One damped natural-gradient step with backtracking (variational_inference.py:
        762-802).  Returns (orig_obj, new_obj)."""
        if self._hyper is None:
            raise RuntimeError('nat_grad_vi_delta must always be set prior to running '
                               '_update_beta')
        self._lsr = lsr
        while True:
            # the step the search would try next rides along (see _launch_trial)
            new_obj, totals = self._trial(1. / L[idx], 1. / (L[idx] * lsr))
            if self._log_info:
                logging.info('...Old objective = %f, new objective = %f', orig_obj, new_obj)
            # scalar arithmetic on Python floats: np.isclose & co cost ~20 us per call, which is
            # visible next to a 150 us evaluation on an 8-GPU shard
            accepted = new_obj >= orig_obj - REL_TOL * abs(orig_obj) - ABS_TOL
            ahead, self._ahead = self._ahead, None
            if ahead is not None:
                # a stage was queued behind this trial, predicated on the device's own accept
                # test (same arithmetic: the two can only differ through the convergence veto)
                ran = bool(self._pending_flag)
                if ran and not accepted:
                    raise RuntimeError('device and host line-search decisions disagree')
                if ran:
                    self._mine = ahead
                    self.n_stages_ahead += 1
                else:
                    self.engine.spec_restore()      # its kernels exited; undo its index flips
                    self.n_stages_skipped += 1
            if accepted:
                if L[idx] > L_MAX and not np.isclose(orig_obj, new_obj):
                    raise RuntimeError('Encountered a numerical error.')
                self._accept(self._candidate, new_obj, totals,
                             already_flipped=self._mine is not None)
                return orig_obj, new_obj
            if L[idx] > L_MAX:
                if not np.isclose(orig_obj, new_obj):
                    raise RuntimeError('Encountered a numerical error.')
                return orig_obj, orig_obj
            self._L_rejected = L[idx]           # the step 1/L was too long here (see _may_look_ahead)
            L[idx] *= lsr

    def _update_hyper_delta(self, orig_obj, with_diff=False, next_step=None, next_alt=None):
        """Closed-form M-step for the mixture weights (variational_inference.py:825-860), all on
        the device: responsibility sums (already all-reduced if they came with the accepted beta
        trial) -> new hyper_delta and its table -> re-evaluation, then ONE download.
        The update is unconditional in the reference, so it is accepted before the download --
        and because it is unconditional, the NEXT sweep's first beta trial (step `next_step`,
        known from L alone) can be queued right behind it and fetched in the same round trip."""
        eng = self.engine
        L = eng.layout
        mine, self._mine = self._mine, None
        look = self._may_look_ahead(with_diff, next_step)
        lo = L.dsum.start if with_diff else L.totals.start
        flag = None
        if mine is None:
            if self._cur_sums is None:
                # no accepted beta step since the last evaluation (line search gave up or
                # resumed state): compute the statistic of the current state now
                sums = eng.delta_sums()
                if self.comm.active:
                    self.comm.allreduce_inplace(sums)
            # otherwise the sums of the accepted trial are still in the result vector
            # (all-reduced)
            self._queue_mstep_stage(with_diff, next_step, next_alt=next_alt)
            if look:
                out = 0
                info = self._look
                self._queue_decision(lo, self._veto, out, from_state=False,
                                     running=info['running'], ends=info['ends_next'],
                                     delta_beta=info['delta_beta'], before=orig_obj)
        else:
            # this sweep's stage was queued ahead of its decision and has run: M-step,
            # re-evaluation, statistics, next trial, that trial's decision, result copy
            if not (with_diff and next_step == mine['step']):
                raise RuntimeError('a stage queued ahead does not match the sweep it belongs to')
            out = mine['out']
        if look:
            # queue the NEXT sweep's stage behind the decision just queued, then collect this
            # sweep's results: the device never waits for the host on an accepted trial
            step_after = self._look['step_after']
            eng.spec_save()
            eng.set_predicate(out)
            eng.accept(True)
            self._queue_mstep_stage(True, step_after, launch_only=True,
                                    next_alt=self._look['alt_after'])
            self._queue_decision(lo, self._veto_next, 1 - out, from_state=True,
                                 ends=self._look['ends_after'])
            eng.set_predicate(None)
            self._ahead = {'pred': out, 'out': 1 - out, 'step': step_after}
        alt = next_alt if self._two_step else None
        if look or mine is not None:
            host, flags = eng.fetch_end(out)
            flag = flags[out]
            self._pending = {'step': next_step, 'alt_step': alt, 'host': host, 'flag': flag}
        elif next_step is not None:
            host = self._fetch(lo, L.sums.stop, with_max=with_diff)
            self._pending = {'step': next_step, 'alt_step': alt, 'host': host}
        else:
            host = self._fetch(lo, L.totals.stop, with_max=with_diff)
        totals = host[L.totals]
        self._install_hyper(host[L.hyper].reshape(self.num_annotations, self.num_mix))
        self._last_diff = np.concatenate([host[L.dsum], host[L.dmax]]) if with_diff else None
        new_obj = self._objective_from(totals)
        self.n_evaluations += 1
        self.n_products += 1
        self._objective, self._totals = new_obj, totals
        self._cur_sums = self._trial_sums = None
        self._version += 1
        if self._log_info:
            logging.info('...Old objective = %f, new objective = %f', orig_obj, new_obj)
        return orig_obj, new_obj

    def _queue_mstep_stage(self, with_diff, next_step, launch_only=False, next_alt=None):
        """The device work of one M-step: new hyper_delta from the sums in the result vector,
        re-evaluation, convergence statistics and the next sweep's first beta trial."""
        eng = self.engine
        eng.mstep()
        eng.eval(diff=with_diff)        # the convergence statistics ride in the same pass
        eng.accept(False)
        if next_step is not None:
            self._launch_trial(next_step, next_alt, keep_sums=launch_only)

    def _queue_decision(self, lo, veto, out_slot, from_state, ends, running=None, delta_beta=0.0,
                        before=0.0):
        """All-reduce what ranks must agree on, take the line-search decision of the trial
        just queued on the device (flag `out_slot`) and start copying the results out."""
        eng = self.engine
        if self.comm.active:
            L = eng.layout
            view = self._views.get((lo, L.sums.stop))
            if view is None:
                view = self._views[(lo, L.sums.stop)] = eng.results[lo:L.sums.stop]
            self.comm.allreduce_inplace(view)
        if self._half_rank_log_tau is None:
            self._half_rank_log_tau = np.array(
                [0.5 * self.ld_ranks[p] * math.log(self.error_scaling[p])
                 for p in range(self.num_pops)])
        # with from_state the device takes the sweep's ELBO change and the running value from
        # what the previous decision left there (the host does not know them yet)
        eng.decide(self.chi_stat, self._half_rank_log_tau, REL_TOL, ABS_TOL, veto, out_slot,
                   from_state=from_state, running=running if not from_state else 0.0,
                   loop_ends_anyway=ends, delta_beta=delta_beta, obj_before_mstep=before,
                   snapshot=True)
        eng.fetch_begin(out_slot, snapshot=True)

    def _may_look_ahead(self, with_diff, next_step):
        """Queue the next sweep's stage ahead of its decision?  The device can tell a standard
        sweep (first beta trial accepted, inner loop ends after it) on its own; not with an
        error-scaling update or per-sweep logging in the sweep, and only if the caller has
        promised that another sweep follows."""
        if not (self._look_ok and with_diff and next_step is not None and not self.scale_se
                and not self._verbose and hasattr(self.engine, 'decide')
                and 1. / next_step * 1.25 < L_MAX
                and os.environ.get('VILMA_LOOKAHEAD', '1') != '0'):
            return False
        # A stage queued ahead of a trial that is then rejected costs more (a dozen empty
        # launches, then the slow path) than it saves when the trial is accepted, and rejections
        # are predictable: L shrinks by 1.25 per sweep until the step 1/L is too long again, which
        # happens near the L of the previous rejection.  Do not run ahead of such a trial.
        rejected = getattr(self, '_L_rejected', None)
        return rejected is None or 1. / next_step > LOOKAHEAD_MARGIN * rejected

    def _drop_ahead(self):
        """Forget a stage queued ahead whose decision is not going to be looked at through the
        normal path (its trial is not the one wanted): wait for it and undo it if it did not
        run."""
        ahead, self._ahead = self._ahead, None
        if ahead is None:
            return
        _, flags = self.engine.fetch_end(ahead['out'])
        if flags[ahead['pred']]:
            raise RuntimeError('a stage queued ahead ran although its sweep was abandoned')
        self.engine.spec_restore()
        self.n_stages_skipped += 1

    def _update_error_scaling(self):
        """EM update of the SE scaling (variational_inference.py:472-486, 735-738) from the
        sums of the current state; the sigma-dependent constants follow tau inside the kernels."""
        P = self.num_pops
        t = self._totals
        lin, var, quad = t[:P], t[P:2 * P], t[2 * P:3 * P]
        self.error_scaling = (self.chi_stat - 2 * lin + quad + var) / self.ld_ranks
        self._half_rank_log_tau = None
        self.engine.set_tau(self.error_scaling)

    def _nat_grad_step(self, L, line_search_rate, running_elbo_delta=None):
        """variational_inference.py:419-450 with the redundant re-evaluations removed: the
        objective of the state a parameter-set update starts from is the one already computed
        when that state was accepted."""
        conv_tol = float('inf') if running_elbo_delta is None else 0.1 * running_elbo_delta
        delta_sum = 0
        # ---- paramset 0: variational family of beta
        orig_obj = self._objective
        for _ in range(MAX_NUM_ITERS):
            L[0] = max([1., L[0] / 1.25])
            if self._log_info:
                logging.info('...Updating paramset %d, L=%f', 0, L[0])
            orig_obj, new_obj = self._update_beta(L, 0, line_search_rate, orig_obj)
            delta_sum += new_obj - orig_obj
            # == np.isclose(new_obj - orig_obj, 0, atol=conv_tol, rtol=0) for finite objectives
            if abs(new_obj - orig_obj) <= conv_tol or L[0] == 1 or L[0] > L_MAX:
                break
            if self._mine is not None:
                raise RuntimeError('device and host disagree on the end of the beta loop')
            orig_obj = new_obj
        # ---- paramset 1: mixture weights (L[1] stays 1, so exactly one pass)
        L[1] = max([1., L[1] / 1.25])
        if self._log_info:
            logging.info('...Updating paramset %d, L=%f', 1, L[1])
        # without --learn-scaling this is the last evaluation of the sweep: piggy-back the
        # convergence statistics on its download
        last = not self.scale_se
        next_L = max([1., L[0] / 1.25])          # L of the next sweep's first trial
        spec = 1. / next_L if (self._speculate and last) else None
        after_L = max([1., next_L / 1.25])       # ... and of the sweep after, if that one is accepted
        lsr = line_search_rate
        self._look = {'delta_beta': delta_sum, 'running': running_elbo_delta,
                      'ends_next': bool(next_L == 1), 'ends_after': bool(after_L == 1),
                      'step_after': 1. / after_L, 'alt_after': 1. / (after_L * lsr)}
        orig_obj, new_obj = self._update_hyper_delta(
            self._objective, with_diff=self._want_diff and last, next_step=spec,
            next_alt=None if spec is None else 1. / (next_L * lsr))
        delta_sum += new_obj - orig_obj
        # ---- paramset 2: annotations -- nothing to do in this scheme (:862-866)
        L[2] = max([1., L[2] / 1.25])
        if self._log_info:
            logging.info('...Updating paramset %d, L=%f', 2, L[2])
        if self.scale_se and delta_sum < EM_TOL:
            orig_obj = self._objective
            self._update_error_scaling()
            new_obj, totals = self._evaluate()
            self._accept(False, new_obj, totals)
            delta_sum += new_obj - orig_obj
            logging.info('...Updating error_scaling, old ELBo=%f, new ELBo=%f', orig_obj, new_obj)
        return L, delta_sum

    def _optimize_step(self, params, L, curr_elbo, line_search_rate=1.25,
                       running_elbo_delta=None):
        """variational_inference.py:396-410."""
        if hasattr(self.engine, 'refresh_stream'):
            self.engine.refresh_stream()
        self._upload(params)
        self._require_fixed_point('_optimize_step')
        # one check per sweep instead of one per message (each costs ~1.3 us even when disabled,
        # on the path between a decision and the next launch)
        self._log_info = logging.getLogger().isEnabledFor(logging.INFO)
        if self._log_info:
            logging.info('Current ELBO = %f and L = %f,%f,%f,%f,%f', curr_elbo, *L[:5])
        L_new, elbo_change = self._nat_grad_step(L, line_search_rate, running_elbo_delta)
        elbo = curr_elbo + elbo_change
        if running_elbo_delta is None:
            running_elbo_delta = elbo_change
        running_elbo_delta *= ELBO_MOMENTUM
        running_elbo_delta += (1 - ELBO_MOMENTUM) * max(elbo_change, 0)
        return self._params(), L_new, elbo, running_elbo_delta

    def _diff_stats(self):
        """Statistics of the posterior-mean change over the sweep just finished (fetched with
        its last evaluation when possible, otherwise computed and fetched now)."""
        if self._last_diff is not None:
            d, self._last_diff = self._last_diff, None
            return d
        self.engine.mean_diff()
        L = self.engine.layout
        host = self._fetch(L.dsum.start, L.dsum.stop, with_max=True)
        return np.concatenate([host[L.dsum], host[L.dmax]])

    def sweep(self, state=None, lookahead=False):
        """One outer iteration as optimize() runs it: _optimize_step + convergence statistics.
        `state` carries (L, elbo, running_elbo_delta) between calls; returns (state, stats).
        lookahead=True promises that sweep() is called again: in the steady state the next
        sweep's M-step stage is then queued ahead of its line-search decision, which the device
        takes itself (the state reported by this call is still the state after THIS sweep; the
        device may already be one sweep further)."""
        if state is None:
            self.engine.snapshot_mean()
            state = {'L': np.ones(5), 'elbo': self._objective, 'running': None}
        self._want_diff, self._last_diff, self._speculate = True, None, True
        self._look_ok, self._veto, self._veto_next = bool(lookahead), False, False
        _, L, elbo, running = self._optimize_step(self._params(), L=state['L'],
                                                   curr_elbo=state['elbo'], line_search_rate=2.,
                                                   running_elbo_delta=state['running'])
        stats = self._diff_stats()
        return {'L': L, 'elbo': elbo, 'running': running}, stats

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
        self._verbose, self._want_diff, self._last_diff = verbose, True, None
        self._speculate = True
        ckp_mean = self.real_posterior_mean(params) if (verbose and self.checkpoint) else None
        while num_its < self.num_its and not converged:
            if self.checkpoint and num_its % self.checkpoint_freq == 0:
                fname = '{}.{}'.format(self.checkpoint_path, num_its)
                dump = self.create_dump_dict(params)
                if self.comm.rank == 0:
                    np.savez(fname, **dump)
                if verbose:
                    ckp_mean = self.real_posterior_mean(params)
            # May the device run ahead of the host through the NEXT sweep's decision?  Only if
            # this sweep cannot end the loop on a criterion the device does not see: the sweep
            # count and the running-ELBO rule (running_s >= MOMENTUM * running_{s-1}, so it
            # stays above the tolerance whenever MOMENTUM * |running| does); "no posterior mean
            # moved" is vetoed on the device itself; checkpoints need the state of their sweep.
            fresh_start = num_its < 10 and loaded_checkpoint is None
            self._veto = not fresh_start
            self._veto_next = not (num_its + 1 < 10 and loaded_checkpoint is None)
            self._look_ok = (num_its + 1 < self.num_its
                             and (fresh_start or (running is not None and
                                                  ELBO_MOMENTUM * abs(running) > ELBO_TOL))
                             and not (self.checkpoint and (num_its + 1) % self.checkpoint_freq == 0))
            params, L, elbo, running = self._optimize_step(
                params, L=L, curr_elbo=elbo, line_search_rate=2., running_elbo_delta=running)
            d = self._diff_stats()
            converged = d[0] == 0
            converged = converged or abs(running) <= ELBO_TOL
            if num_its < 10 and loaded_checkpoint is None:
                converged = False
            if verbose:
                self._dump_info(num_its, d, n_total, params, ckp_mean)
            num_its += 1
        self._look_ok = False
        self._drop_ahead()              # a stage vetoed by convergence: undo its bookkeeping
        self._speculate, self._pending = False, None
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
    the LD storage ('auto' | 'dense' | 'eig'); `_engine_factory` lets tests inject a different
    engine implementation (the default is the HIP engine and there is no CPU fallback)."""

    def __init__(self, marginal_effects=None, std_errs=None, ld_mats=None, annotations=None,
                 mixture_covs=None, checkpoint=True, checkpoint_freq=5, scaled=False,
                 scale_se=False, output='vilma_output', gwas_N=None, init_hg=None,
                 num_its=None, form='auto', _engine_factory=None, _comm=None):
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
        if _engine_factory is None:
            from .engine import HipEngine
            _engine_factory = HipEngine
        self.engine = _engine_factory(P, n_loc, M, self.num_annotations)

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
        ridge = ld_device.ridge_start(self.engine, rmle, reg, diag_loc)   # rhs = adj * se
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
        if which == 'vi_mu':
            return self.comm.gather_snps(self.engine.get_mu(), self._snps, self.num_loci)
        self._pending = None        # vilma_get_delta uses the trial vi_mu buffer as scratch
        delta = self.engine.get_delta()
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
        """[M,P,P,N], computed on demand for outputs (variational_inference.py:712-724)."""
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

