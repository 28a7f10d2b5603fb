"""A CPU engine with the HipEngine interface, built on the oracle -- TEST ONLY.

Lets the host driver (vilma_amd.variational_inference.MultiPopVI) and the sharding / all-reduce
protocol be exercised without a GPU (`-m "not gpu"`, gloo world_size 2).  The product never
imports this: the default engine is the HIP one and there is no CPU fallback.
"""
import numpy as np
import torch

from oracle import numerics as nm
from oracle.ldop import BlockDiagonalLD


class _Block:
    def __init__(self, spec):
        if spec[0] == 'dense':
            self.R = np.asarray(spec[1])
            self.n = self.R.shape[0]
            self.dot = lambda x: self.R @ x
        else:
            U, s = np.asarray(spec[1]), np.asarray(spec[2])
            self.n = U.shape[0]
            self.dot = lambda x: U @ (s * (U.T @ x))
        self.shape = (self.n, self.n)


class OracleEngine:
    def __init__(self, P, N, M, A):
        self.P, self.N, self.M, self.A = P, N, M, A
        self.tau = np.ones(P)
        self.ld = [None] * P
        self.cur = None          # dict(mean, var, z, linked, delta)
        self.trial_state = None
        self.mu = None
        self.mu_trial = None
        self.snap = None
        self._c = None
        from vilma_amd.engine import ResultLayout
        self.n_totals = 3 * P + 2
        self.layout = ResultLayout(self.n_totals, A * M)
        L = self.layout
        self.results = torch.zeros(L.size, dtype=torch.float64)
        self._dsum, self._totals = self.results[L.dsum], self.results[L.totals]
        self._ttotals = self.results[L.ttotals]
        self._ttotals_b = self.results[L.ttotals_b]
        self._sums, self._dmax = self.results[L.sums], self.results[L.dmax]
        self._hyper_t = self.results[L.hyper]
        self._last_trial_kind = None

    # ---- static data
    def set_snp_data(self, adj, se, sld, scalings, annot):
        self.adj, self.se, self.sld = np.array(adj), np.array(se), np.array(sld)
        self.scal, self.annot = np.array(scalings), np.array(annot, dtype=np.int64)
        self._c = None

    def set_mixture(self, prec, log_det):
        self.prec = np.array(prec).reshape(self.M, self.P, self.P)
        self.log_det = np.array(log_det)
        self._c = None

    def set_tau(self, tau):
        self.tau = np.array(tau, dtype=float)
        self._c = None

    def set_hyper(self, hyper):
        self.hyper = np.array(hyper).reshape(self.A, self.M)

    def set_annotation_counts(self, counts):
        self.counts = np.array(counts, dtype=float)

    def mstep(self, sums=None):
        sums = self._sums if sums is None else sums
        h = np.maximum(sums.numpy().reshape(self.A, self.M)
                       / (self.counts.reshape((-1, 1)) + 1e-100), 1e-100)
        h /= h.sum(axis=1, keepdims=True)
        self.hyper = h
        self._hyper_t.copy_(torch.as_tensor(h.ravel()))
        return self._hyper_t

    def load_ld(self, cohort, blocks, perm, n_ld):
        perm = np.asarray(perm)
        self.ld[cohort] = BlockDiagonalLD([_Block(b) for b in blocks], perm=perm,
                                          missing=perm[n_ld:])

    def ld_matvec(self, x, cohort=-1):
        x = np.asarray(x).reshape(self.P, self.N)
        out = np.zeros_like(x)
        for p in range(self.P):
            if cohort < 0 or cohort == p:
                out[p] = self.ld[p].dot(x[p])
        return out

    def ld_bytes(self):
        return 0, 0

    # ---- sigma-dependent constants (variational_inference.py:712-733)
    def _consts(self):
        if self._c is None:
            lam = np.zeros((self.M, self.P, self.P, self.N))
            idx = np.arange(self.P)
            lam[:, idx, idx, :] = self.sld / self.tau[:, None]
            lam += self.prec[:, :, :, None]
            sigma = nm.vi_sigma_inv(lam)
            logdet = nm.vi_sigma_log_det(sigma)
            matches = np.einsum('kpq,kqpi->ik', self.prec, sigma)
            self._c = dict(sigma=sigma, nat_sigma=-0.5 * lam, logdet=logdet,
                           summary=self.log_det - logdet.T + matches)
        return self._c

    def _delta(self, mu, nat_mu=None):
        c = self._consts()
        if nat_mu is None:
            nat_mu = nm.fast_nat_inner_product_m2(mu, c['nat_sigma'])
        grad = nm.fast_vi_delta_grad(self.hyper, self.log_det, self.annot)
        return nm.fast_invert_nat_vi_delta(mu, nat_mu, np.copy(c['logdet'].T), grad)

    def _moments(self, mu, delta, out=None):
        c = self._consts()
        mean = nm.fast_posterior_mean(mu, delta)
        var = nm.fast_pmv(mean, mu, delta, np.einsum('kppi->kpi', c['sigma']))
        z = mean / self.se
        linked = self.ld_matvec(z)
        st = dict(mean=mean, var=var, z=z, linked=linked, delta=delta)
        totals = np.concatenate([
            (mean * self.adj).sum(axis=1), (self.sld * var).sum(axis=1),
            (linked * z).sum(axis=1),
            [nm.fast_delta_kl(delta, self.hyper, self.annot) + nm.fast_beta_kl(c['summary'], delta),
             nm.fast_inner_product_comp(mu, self.prec[:, :, :, None], delta)]])
        out = self._totals if out is None else out
        out.copy_(torch.as_tensor(totals))
        return st, out

    # ---- state
    def set_mu(self, vi_mu):
        self.mu = np.array(vi_mu)
        self.cur = None

    def get_mu(self):
        return np.copy(self.mu)

    def get_delta(self):
        return np.copy(self.cur['delta'])

    def get_moments(self):
        return np.copy(self.cur['mean']), np.copy(self.cur['var'])

    def init_state(self, fake_mu):
        """_initialize's per-SNP part (variational_inference.py:658-692), restated with numpy."""
        c = self._consts()
        f = np.asarray(fake_mu)
        matches = np.einsum('kpq,kqpi->ik', self.prec, c['sigma'])
        probs = np.einsum('pi,oi,kpo->ik', 1.6 * f, 1.6 * f, self.prec) + matches - self.log_det
        probs = np.exp(-0.5 * (probs - probs.min(axis=1, keepdims=True)))
        delta = np.maximum(probs / probs.sum(axis=1, keepdims=True), nm.EPSILON)
        avg = np.einsum('kpqi,ik->ipq', c['sigma'], delta)
        nat = np.einsum('pi,iqp->qi', f, np.linalg.inv(avg))
        self.mu = np.einsum('kqpi,pi->kqi', c['sigma'], nat)
        self.cur = None
        self._sums.copy_(torch.as_tensor(nm.sum_annotations(delta, self.annot, self.A).ravel()))
        return self._sums

    def eval_given_delta(self, vi_delta):
        derived = self.cur['delta']
        keep = self._totals.clone()
        self.given_state, totals = self._moments(self.mu, np.asarray(vi_delta))
        out = torch.cat([totals.clone(), torch.as_tensor([np.abs(vi_delta - derived).max()])])
        self._totals.copy_(keep)
        return out

    def get_trial_moments(self):
        return np.copy(self.given_state['mean']), np.copy(self.given_state['var'])

    # ---- evaluations
    def eval(self, diff=False):
        self.mu_trial = None
        self.trial_state, totals = self._moments(self.mu, self._delta(self.mu))
        if diff:
            self._diff_against_snapshot(self.trial_state['mean'])
        return totals

    def trial(self, step):
        c = self._consts()
        cur = self.cur
        old_nat = nm.fast_nat_inner_product_m2(self.mu, c['nat_sigma'])
        linked = nm.fast_linked_ests(cur['linked'], self.se, cur['mean'], self.sld)
        grad = np.broadcast_to(((self.adj - linked) / self.tau[:, None])[None], self.mu.shape)
        nat_mu = nm.sum_betas(old_nat, grad, step)
        new_mu = nm.fast_nat_inner_product(nat_mu, c['sigma'])
        self.mu_trial = new_mu
        self.trial_state, totals = self._moments(new_mu, self._delta(new_mu, nat_mu),
                                                 out=self._ttotals)
        return totals

    def trial2(self, step_a, step_b):
        self.trial(step_b)
        self._ttotals_b.copy_(self._ttotals)
        self.mu_trial_b, self.trial_state_b = self.mu_trial, self.trial_state
        self.trial(step_a)
        return self._ttotals, self._ttotals_b

    def accept(self, take_mu):
        if int(take_mu) == 2:
            self.mu, self.cur = self.mu_trial_b, self.trial_state_b
            return
        if take_mu:
            self.mu = self.mu_trial
        self.cur = self.trial_state

    def delta_sums(self, which=0):
        st = self.cur if which == 0 else (self.trial_state_b if which == 3 else self.trial_state)
        self._sums.copy_(torch.as_tensor(nm.sum_annotations(st['delta'], self.annot,
                                                            self.A).ravel()))
        return self._sums

    def snapshot_mean(self):
        self.snap = self.cur['mean'] * self.scal

    def mean_diff(self):
        return self._diff_against_snapshot(self.cur['mean'])

    def _diff_against_snapshot(self, mean):
        new = mean * self.scal
        old = self.snap
        df = np.abs(new - old)
        out = np.array([np.sum(df > 1e-6 + 1e-6 * np.abs(old)), df.sum(), (df ** 2).sum(),
                        np.abs(new).max(), df.max(), np.abs((new - old) / (old + 1e-100)).max()],
                       dtype=float)
        self.snap = new
        self._dsum.copy_(torch.as_tensor(out[:3]))
        self._dmax.copy_(torch.as_tensor(out[3:]))
        return torch.as_tensor(out)

    def fetch(self):
        return self.results.numpy().copy()

    def close(self):
        pass

    # ---- measurement hooks of HipEngine: nothing to measure here (bench.py's CPU rehearsal of
    # the multi-rank launch line, tests/bench_rehearsal.py, still walks through them)
    PROF_KINDS = ('ld_sym_kernel', 'ld_eig_fused_kernel', 'ld_sym_kernel_two_rhs',
                  'snp_pass_eval', 'snp_pass_trial', 'snp_pass_trial2',
                  'sums_pass', 'sums_pass_store', 'snp_pass_trial_lazy', 'snp_pass_trial2_lazy')

    def prof_enable(self, on=True, every=1):
        pass

    def prof_read(self, reset=True):
        return {k: (0.0, 0) for k in self.PROF_KINDS}

    def stream_store(self, passes=5):
        raise RuntimeError('no LD store on a device')

    def comm_check(self):
        """The number of ranks an all-reduce of a one per rank reaches."""
        if getattr(self, 'comm', None) is None or not self.comm.active:
            return 1
        return int(round(float(self.comm.allreduce_np(np.array([1.0]))[0])))

    # ---- the sweep behind one call: the engine-level interface of HipEngine (vilma_sweep & co.),
    # with the reference's line search (variational_inference.py:396-450, 762-802, 825-860)
    # restated in Python on the oracle-backed evaluations above.  TEST ONLY: the product's sweep
    # is csrc/sweep.hip.
    L_MAX, REL_TOL, ABS_TOL, EM_TOL, MAX_NUM_ITERS = 1e12, 1e-6, 1e-6, 10, 20

    def set_fit_constants(self, chi_stat, ld_ranks, scale_se):
        self.chi, self.ranks = np.array(chi_stat, dtype=float), np.array(ld_ranks, dtype=float)
        self.scale_se = bool(scale_se)

    def bind_comm(self, comm):
        self.comm = comm
        self.collective = 'torch.distributed (test engine)'

    def _reduce(self, sl, op='sum'):
        if getattr(self, 'comm', None) is not None and self.comm.active:
            self.comm.allreduce_inplace(self.results[sl], op=op)

    def _objective_from(self, t):
        P = self.P
        t = np.asarray(t).tolist()
        lik = 0.0
        for p in range(P):
            lik += ((-0.5 * (t[P + p] + t[2 * P + p]) + t[p] - 0.5 * self.chi[p]) / self.tau[p]
                    - 0.5 * self.ranks[p] * np.log(self.tau[p]))
        return lik - (t[3 * P] + t[3 * P + 1])

    def _evaluate(self):
        L = self.layout
        self.eval()
        self._reduce(L.totals)
        totals = self.results[L.totals].numpy().copy()
        return self._objective_from(totals), totals

    def set_state(self, vi_mu, hyper, tau=None):
        if tau is not None:
            self.set_tau(tau)
        self.set_hyper(hyper)
        if vi_mu is not None:
            self.set_mu(vi_mu)
        self._obj, self._tot = self._evaluate()
        self.accept(False)
        self._cur_sums = False
        return self._obj

    def initialize(self, fake_mu):
        self.init_state(fake_mu)
        self._reduce(self.layout.sums)
        sums = self._sums.numpy().reshape(self.A, self.M)
        hyper = sums + 1.
        hyper /= hyper.sum(axis=1, keepdims=True)
        return self.set_state(None, np.maximum(hyper, 1e-100))

    def get_hyper(self):
        return np.array(self.hyper).reshape(self.A, self.M)

    def get_tau(self):
        return np.array(self.tau)

    def elbo(self):
        return self._obj

    def drain(self):
        pass

    def posterior(self):
        return self.cur['mean'] * self.scal, self.cur['var'] * self.scal ** 2

    def sweep(self, L, elbo, running, line_search_rate=2., flags=0):
        from types import SimpleNamespace
        lay = self.layout
        st = SimpleNamespace(n_evaluations=0, n_trials=0, n_products=0, ran_ahead=0,
                             skipped_ahead=0, n_events=0, events=[], diff_sum=[0.] * 3,
                             diff_max=[0.] * 3)
        want_diff, verbose = bool(flags & 1), bool(flags & 16)
        conv_tol = float('inf') if running is None else 0.1 * running
        delta_sum = 0.
        orig = self._obj
        for _ in range(self.MAX_NUM_ITERS):
            L[0] = max(1., L[0] / 1.25)
            # ---- _update_beta: backtracking on L[0]
            alt = None
            while True:
                step = 1. / L[0]
                if alt is not None and alt[0] == step:
                    new, totals, cand, with_sums = alt[1], alt[2], 2, False
                    alt = None
                else:
                    self.trial2(step, 1. / (L[0] * line_search_rate))
                    self.delta_sums(1)
                    self._reduce(slice(lay.ttotals.start, lay.sums.stop))
                    host = self.results.numpy().copy()
                    totals, cand, with_sums = host[lay.ttotals], 1, True
                    new = self._objective_from(totals)
                    tb = host[lay.ttotals_b]
                    alt = (1. / (L[0] * line_search_rate), self._objective_from(tb), tb)
                    st.n_products += 1
                st.n_evaluations += 1
                st.n_trials += 1
                if new >= orig - self.REL_TOL * abs(orig) - self.ABS_TOL:
                    if L[0] > self.L_MAX and not np.isclose(orig, new):
                        raise RuntimeError('Encountered a numerical error.')
                    self.accept(cand)
                    self._obj, self._tot, self._cur_sums = new, totals, with_sums
                    break
                if L[0] > self.L_MAX:
                    if not np.isclose(orig, new):
                        raise RuntimeError('Encountered a numerical error.')
                    new = orig
                    break
                L[0] *= line_search_rate
            delta_sum += new - orig
            if abs(new - orig) <= conv_tol or L[0] == 1 or L[0] > self.L_MAX:
                break
            orig = new
        # ---- _update_hyper_delta
        L[1] = max(1., L[1] / 1.25)
        last = not self.scale_se
        if not self._cur_sums:
            self.delta_sums(0)
            self._reduce(lay.sums)
        self.mstep()
        self.eval(diff=want_diff and last)
        self.accept(False)
        self._reduce(slice(lay.dsum.start if (want_diff and last) else lay.totals.start,
                           lay.totals.stop))
        if want_diff and last and verbose:
            self._reduce(lay.dmax, op='max')
        host = self.results.numpy().copy()
        orig, self._tot = self._obj, host[lay.totals]
        self._obj = self._objective_from(self._tot)
        self._cur_sums = False
        st.n_evaluations += 1
        st.n_products += 1
        delta_sum += self._obj - orig
        L[2] = max(1., L[2] / 1.25)
        if self.scale_se and delta_sum < self.EM_TOL:
            P, t = self.P, self._tot
            orig = self._obj
            self.set_tau((self.chi - 2 * t[:P] + t[2 * P:3 * P] + t[P:2 * P]) / self.ranks)
            self._obj, self._tot = self._evaluate()
            self.accept(False)
            st.n_evaluations += 1
            st.n_products += 1
            delta_sum += self._obj - orig
        if want_diff:
            if not last:
                self.mean_diff()
                self._reduce(lay.dsum)
                if verbose:
                    self._reduce(lay.dmax, op='max')
                host = self.results.numpy().copy()
            st.diff_sum, st.diff_max = list(host[lay.dsum]), list(host[lay.dmax])
        if running is None:
            running = delta_sum
        running = 0.5 * running + 0.5 * max(delta_sum, 0)
        st.error_scaling = list(self.tau) + [1.] * (8 - self.P)
        st.objective = self._obj
        return elbo + delta_sum, running, st
