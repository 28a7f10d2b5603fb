"""Oracle restatement of the reference's VI engine (variational_inference.py).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  `MultiPopVIOracle` follows
`VIScheme` + `MultiPopVI` (variational_inference.py:27-889) with the reference's
operation schedule: every objective evaluation does one LD matvec per cohort and the same
sequence of whole-array passes, so that timing it on the GPU box's host is a fair stand-in
for the reference CPU path (bench.py `cpu_baseline`, kind "port").

Counters (`n_matvec`, `n_objective`) let tests compare the schedule with the golden
trajectories recorded from the reference.
"""
import logging

import numpy as np

from . import numerics as nm
from .ldop import BlockDiagonalLD

L_MAX = 1e12            # variational_inference.py:18-24
REL_TOL = 1e-6
ABS_TOL = 1e-6
ELBO_TOL = 0.1
EM_TOL = 10
ELBO_MOMENTUM = 0.5
MAX_NUM_ITERS = 20


class MultiPopVIOracle:
    def __init__(self, marginal_effects=None, std_errs=None, ld_mats=None, annotations=None,
                 mixture_covs=None, checkpoint=True, checkpoint_freq=5, scaled=False,
                 scale_se=False, output='vilma_output', gwas_N=None, init_hg=None,
                 num_its=None):
        # ---- MultiPopVI.__init__ (variational_inference.py:599-630) ----
        P = marginal_effects.shape[0]
        for mc in mixture_covs:
            if mc.shape != (P, P):
                raise ValueError('Mixture component has a covariance matrix of the wrong '
                                 'shape.')
        if not np.all(np.linalg.slogdet(mixture_covs)[0] == 1):
            raise ValueError('Mixture component has a non-positive definite covariance '
                             'matrix.')
        self.num_mix = len(mixture_covs)
        # ---- VIScheme.__init__ (variational_inference.py:96-259) ----
        for name, val in (('init_hg', init_hg), ('gwas_N', gwas_N),
                          ('marginal_effects', marginal_effects), ('std_errs', std_errs),
                          ('ld_mats', ld_mats), ('annotations', annotations),
                          ('num_its', num_its)):
            if val is None:
                raise ValueError('%s must be specified' % name)
        if not np.all(np.isfinite(marginal_effects)):
            raise ValueError('Encountered an infinite or NaN value in the GWAS effect size '
                             'estimates')
        if not np.all(np.isfinite(std_errs)):
            raise ValueError('Encountered an infinity or NaN value in the GWAS standard '
                             'errors')
        self.scaled, self.scale_se = scaled, scale_se
        self.error_scaling = np.ones(P)
        self.checkpoint, self.checkpoint_freq = checkpoint, checkpoint_freq
        self.checkpoint_path = '%s-checkpoint' % output
        self.num_pops, self.num_loci = marginal_effects.shape
        if len(ld_mats) != P:
            raise ValueError('Fewer LD matrices than populations.')
        for ld in ld_mats:
            if not isinstance(ld, BlockDiagonalLD):
                raise ValueError('LD Matrices must be of type BlockDiagonalLD.')
            if ld.shape != (self.num_loci, self.num_loci):
                raise ValueError('LD matrix shape does not match GWAS marginal effect size '
                                 'shape.')
        self.ld_diags = np.stack([ld.diag() for ld in ld_mats])
        if not np.allclose(annotations.sum(axis=1), 1):
            raise ValueError('Some SNPs are either missing annotations or have more than '
                             'one annotation.')
        if annotations.shape[0] != self.num_loci:
            raise ValueError('annotations dimension does not match GWAS marginal effect '
                             'size shape.')
        self.num_annotations = annotations.shape[1]
        if scaled:                                               # :205-214
            self.marginal_effects = marginal_effects / (std_errs + nm.EPSILON)
            self.std_errs = np.ones_like(std_errs)
            self.scalings = std_errs + nm.EPSILON
        else:
            self.marginal_effects = np.copy(marginal_effects)
            self.std_errs = np.copy(std_errs)
            self.scalings = np.ones_like(std_errs)
        self.scaled_ld_diags = self.std_errs ** -2 * self.ld_diags
        self.ld_mats = ld_mats
        self.annotations = np.copy(np.where(annotations)[1])
        self.annotation_counts = annotations.sum(axis=0)
        self.init_hg, self.gwas_N, self.num_its = init_hg, gwas_N, num_its
        self.param_names = ['vi_mu', 'vi_delta', 'hyper_delta']
        self.n_matvec = 0
        self.n_objective = 0

        self.adj_marginal_effects = np.zeros_like(self.marginal_effects)
        self.chi_stat = np.zeros(P)
        self.ld_ranks = np.zeros(P)
        self.inverse_betas = np.zeros_like(self.marginal_effects)
        for p in range(P):                                        # :236-252
            z = self.marginal_effects[p] / self.std_errs[p]
            mle = ld_mats[p].inverse_dot(z)
            self.chi_stat[p] = z.dot(mle)
            adj = ld_mats[p].dot(mle) / self.std_errs[p]
            self.adj_marginal_effects[p] = adj
            self.ld_ranks[p] = ld_mats[p].get_rank()
            prior = 2 * gwas_N[p] * init_hg[p] / (self.std_errs[p] ** -2).sum()
            ridge = ld_mats[p].ridge_inverse_dot(adj * self.std_errs[p],
                                                 self.std_errs[p] ** 2 / prior)
            self.inverse_betas[p] = ridge * self.std_errs[p]
        if not np.allclose(self.adj_marginal_effects[np.isclose(self.ld_diags, 0)], 0):
            raise ValueError('Some SNPs that are missing in the LD matrix are not being '
                             'treated as missing.')

        covs = np.array(mixture_covs)[:, :, :, None]              # :622-626
        self.mixture_prec = nm.vi_sigma_inv(covs)
        self.log_det = np.copy(nm.vi_sigma_log_det(covs)[:, 0])
        self._set_vi_sigma()
        self.nat_grad_vi_delta = None

    # ------------------------------------------------------------------ constants
    def _set_vi_sigma(self):
        """variational_inference.py:712-733."""
        M, P, N = self.num_mix, self.num_pops, self.num_loci
        lam = np.zeros((M, P, P, N))
        idx = np.arange(P)
        lam[:, idx, idx, :] = (self.std_errs ** -2 * self.ld_diags
                               / self.error_scaling.reshape((-1, 1)))
        lam += self.mixture_prec
        self.vi_sigma = nm.vi_sigma_inv(lam)
        self.nat_sigma = -0.5 * lam
        self.vi_sigma_log_det = nm.vi_sigma_log_det(self.vi_sigma)
        self.vi_sigma_matches = np.einsum('kpq,kqpi->ik', self.mixture_prec[:, :, :, 0],
                                          self.vi_sigma)
        self.sigma_summary = self.log_det - self.vi_sigma_log_det.T + self.vi_sigma_matches

    def _set_state(self, params):
        """variational_inference.py:702-710."""
        self._set_vi_sigma()
        self.nat_grad_vi_delta = nm.fast_vi_delta_grad(params[2], self.log_det,
                                                       self.annotations)

    # ------------------------------------------------------------------ LD matvec
    def _ld_dot(self, p, vec):
        self.n_matvec += 1
        return self.ld_mats[p].dot(vec)

    # ------------------------------------------------------------------ moments
    def _posterior_mean(self, vi_mu, vi_delta, hyper_delta=None):
        return nm.fast_posterior_mean(vi_mu, vi_delta)

    def _posterior_marginal_variance(self, mean, vi_mu, vi_delta, hyper_delta=None):
        """variational_inference.py:757-760."""
        diag_sigma = np.einsum('kppi->kpi', self.vi_sigma)
        return nm.fast_pmv(mean, vi_mu, vi_delta, diag_sigma)

    def real_posterior_mean(self, vi_mu, vi_delta, hyper_delta=None):
        return self._posterior_mean(vi_mu, vi_delta) * self.scalings

    def real_posterior_variance(self, vi_mu, vi_delta, hyper_delta=None):
        mean = self._posterior_mean(vi_mu, vi_delta)
        return self._posterior_marginal_variance(mean, vi_mu, vi_delta) * self.scalings ** 2

    # ------------------------------------------------------------------ objective
    def _log_likelihood(self, params):
        """variational_inference.py:452-470."""
        mean = self._posterior_mean(*params)
        var = self._posterior_marginal_variance(mean, *params)
        scaled_mu = nm.fast_divide(mean, self.std_errs)
        linked = np.empty_like(mean)
        for p in range(self.num_pops):
            linked[p] = self._ld_dot(p, scaled_mu[p])
        return nm.fast_likelihood(mean, var, scaled_mu, self.scaled_ld_diags, linked,
                                  self.adj_marginal_effects, self.chi_stat, self.ld_ranks,
                                  self.error_scaling)

    def _beta_KL(self, vi_mu, vi_delta, hyper_delta):
        """variational_inference.py:873-885."""
        return (nm.fast_delta_kl(vi_delta, hyper_delta, self.annotations)
                + nm.fast_inner_product_comp(vi_mu, self.mixture_prec, vi_delta)
                + nm.fast_beta_kl(self.sigma_summary, vi_delta))

    def _beta_objective(self, params):
        self.n_objective += 1
        return self._log_likelihood(params) - self._beta_KL(*params)

    def elbo(self, params):
        """variational_inference.py:412-417 (annotation KL is identically 0, :887-889)."""
        self.n_objective += 1
        return self._log_likelihood(params) - self._beta_KL(*params)

    # ------------------------------------------------------------------ updates
    def _nat_to_not_vi_delta(self, params):
        """variational_inference.py:632-641."""
        vi_mu, _, hyper = params
        nat_mu = nm.fast_nat_inner_product_m2(vi_mu, self.nat_sigma)
        delta = nm.fast_invert_nat_vi_delta(vi_mu, nat_mu, np.copy(self.vi_sigma_log_det.T),
                                            self.nat_grad_vi_delta)
        return vi_mu, delta, hyper

    def _nat_grad_beta(self, vi_mu, vi_delta, hyper_delta):
        """variational_inference.py:804-823."""
        mean = self._posterior_mean(vi_mu, vi_delta)
        zs = nm.fast_divide(mean, self.std_errs)
        linked = np.zeros_like(mean)
        for p in range(self.num_pops):
            linked[p] = self._ld_dot(p, zs[p])
        linked = nm.fast_linked_ests(linked, self.std_errs, mean, self.scaled_ld_diags)
        per_pop = (self.adj_marginal_effects - linked) / self.error_scaling[:, None]
        return np.broadcast_to(per_pop[None], vi_mu.shape).copy()

    def _update_beta(self, vi_mu, vi_delta, hyper_delta, orig_obj, L, idx, lsr):
        """variational_inference.py:762-802 -- damped natural-gradient step with
        backtracking on L[idx]."""
        if orig_obj is None:
            orig_obj = self._beta_objective((vi_mu, vi_delta, hyper_delta))
        old_nat_mu = nm.fast_nat_inner_product_m2(vi_mu, self.nat_sigma)
        const_part = np.copy(self.vi_sigma_log_det.T)
        if self.nat_grad_vi_delta is None:
            raise RuntimeError('nat_grad_vi_delta must always be set prior to running '
                               '_update_beta')
        grad = self._nat_grad_beta(vi_mu, vi_delta, hyper_delta)
        while True:
            nat_mu = nm.sum_betas(old_nat_mu, grad, 1. / L[idx])
            new_mu = nm.fast_nat_inner_product(nat_mu, self.vi_sigma)
            new_delta = nm.fast_invert_nat_vi_delta(new_mu, nat_mu, const_part,
                                                    self.nat_grad_vi_delta)
            new_obj = self._beta_objective((new_mu, new_delta, hyper_delta))
            logging.info('...Old objective = %f, new objective = %f', orig_obj, new_obj)
            if new_obj >= orig_obj - REL_TOL * np.abs(orig_obj) - ABS_TOL:
                if L[idx] > L_MAX and not np.isclose(orig_obj, new_obj):
                    raise RuntimeError('Encountered a numerical error.')
                break
            if L[idx] > L_MAX:
                if not np.isclose(orig_obj, new_obj):
                    raise RuntimeError('Encountered a numerical error.')
                return (vi_mu, vi_delta, hyper_delta), L, orig_obj, orig_obj
            L[idx] *= lsr
        return (new_mu, new_delta, hyper_delta), L, orig_obj, new_obj

    def _update_hyper_delta(self, vi_mu, vi_delta, hyper_delta, orig_obj, L, idx, lsr):
        """variational_inference.py:825-860 -- closed-form M-step for mixture weights."""
        if orig_obj is None:
            orig_obj = self.elbo((vi_mu, vi_delta, hyper_delta))
        new_hyper = nm.sum_annotations(vi_delta, self.annotations, self.num_annotations)
        new_hyper = np.maximum(new_hyper / (self.annotation_counts.reshape((-1, 1))
                                            + nm.EPSILON), nm.EPSILON)
        new_hyper /= new_hyper.sum(axis=1, keepdims=True)
        self.nat_grad_vi_delta = nm.fast_vi_delta_grad(new_hyper, self.log_det,
                                                       self.annotations)
        _, new_delta, _ = self._nat_to_not_vi_delta((vi_mu, vi_delta, new_hyper))
        new_obj = self.elbo((vi_mu, new_delta, new_hyper))
        logging.info('...Old objective = %f, new objective = %f', orig_obj, new_obj)
        return (vi_mu, new_delta, new_hyper), L, orig_obj, new_obj

    def _update_annotation(self, vi_mu, vi_delta, hyper_delta, orig_obj, L, idx, lsr):
        """variational_inference.py:862-866 -- a no-op in this scheme."""
        return (vi_mu, vi_delta, hyper_delta), L, 0., 0.

    def _update_error_scaling(self, params):
        """variational_inference.py:472-486 + :735-738."""
        mean = self._posterior_mean(*params)
        var = self._posterior_marginal_variance(mean, *params)
        new = np.zeros_like(self.error_scaling)
        for p in range(self.num_pops):
            zs = mean[p] / self.std_errs[p]
            new[p] = (self.chi_stat[p] - 2 * mean[p].dot(self.adj_marginal_effects[p])
                      + zs.dot(self._ld_dot(p, zs))
                      + (self.ld_diags[p] * var[p] * self.std_errs[p] ** -2).sum()
                      ) / self.ld_ranks[p]
        self.error_scaling = new
        self._set_vi_sigma()

    def _nat_grad_step(self, params, L, line_search_rate, running_elbo_delta=None):
        """variational_inference.py:419-450."""
        conv_tol = float('inf') if running_elbo_delta is None else 0.1 * running_elbo_delta
        delta_sum = 0
        for idx, update in enumerate((self._update_beta, self._update_hyper_delta,
                                      self._update_annotation)):
            orig_obj = None
            for _ in range(MAX_NUM_ITERS):
                L[idx] = max([1., L[idx] / 1.25])
                logging.info('...Updating paramset %d, L=%f', idx, L[idx])
                params, L, orig_obj, new_obj = update(*params, orig_obj, L, idx,
                                                      line_search_rate)
                delta_sum += new_obj - orig_obj
                if (np.isclose(new_obj - orig_obj, 0, atol=conv_tol, rtol=0)
                        or L[idx] == 1 or L[idx] > L_MAX):
                    break
                orig_obj = new_obj
        if self.scale_se and delta_sum < EM_TOL:
            orig_obj = self.elbo(params)
            self._update_error_scaling(params)
            params = self._nat_to_not_vi_delta(params)
            new_obj = self.elbo(params)
            delta_sum += new_obj - orig_obj
            logging.info('...Updating error_scaling, old ELBo=%f, new ELBo=%f',
                         orig_obj, new_obj)
        return params, L, delta_sum

    def _optimize_step(self, params, L, curr_elbo, line_search_rate=1.25,
                       running_elbo_delta=None):
        """variational_inference.py:396-410."""
        logging.info('Current ELBO = %f and L = %f,%f,%f,%f,%f', curr_elbo, *L[:5])
        new_params, L_new, change = self._nat_grad_step(params, L, line_search_rate,
                                                        running_elbo_delta)
        elbo = curr_elbo + change
        if running_elbo_delta is None:
            running_elbo_delta = change
        running_elbo_delta *= ELBO_MOMENTUM
        running_elbo_delta += (1 - ELBO_MOMENTUM) * np.maximum(change, 0)
        return new_params, L_new, elbo, running_elbo_delta

    # ------------------------------------------------------------------ init + driver
    def _initialize(self):
        """variational_inference.py:643-700.  Consumes the legacy global numpy RNG with the
        same call (np.random.normal(loc, scale, size)) as the reference."""
        real_mu = self.inverse_betas
        missing = np.isclose(self.ld_diags, 0)
        fake_mu = np.random.normal(loc=np.copy(real_mu), scale=1e-3 * self.std_errs,
                                   size=real_mu.shape)
        fake_mu[missing] = np.nan
        with np.errstate(all='ignore'):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore', category=RuntimeWarning)
                fill = np.tile(np.nanmean(fake_mu, axis=0), [fake_mu.shape[0], 1])
        fake_mu[missing] = fill[missing]
        fake_mu[np.isnan(fake_mu)] = 0.
        probs = np.einsum('pi,oi,kpo->ik', 1.6 * fake_mu, 1.6 * fake_mu,
                          self.mixture_prec[:, :, :, 0])
        probs += self.vi_sigma_matches
        probs -= self.log_det
        probs = np.exp(-0.5 * (probs - np.min(probs, axis=1, keepdims=True)))
        vi_delta = np.maximum(probs / probs.sum(axis=1, keepdims=True), nm.EPSILON)
        hyper = nm.sum_annotations(vi_delta, self.annotations, self.num_annotations)
        hyper += 1.
        hyper /= np.sum(hyper, axis=1, keepdims=True)
        hyper = np.maximum(hyper, nm.EPSILON)
        nat_vi_delta = nm.fast_vi_delta_grad(hyper, self.log_det, self.annotations)
        avg = np.einsum('kpqi,ik->ipq', self.vi_sigma, vi_delta)
        inv_avg = np.linalg.inv(avg)
        temp_nat_mu = np.einsum('pi,iqp->qi', fake_mu, inv_avg)
        vi_mu = np.einsum('kqpi,pi->kqi', self.vi_sigma, temp_nat_mu)
        self.nat_grad_vi_delta = nat_vi_delta
        _, vi_delta, _ = self._nat_to_not_vi_delta((vi_mu, vi_delta, hyper))
        return vi_mu, vi_delta, hyper

    def create_dump_dict(self, params):
        """variational_inference.py:333-338."""
        dump = dict(zip(self.param_names, params))
        dump['error_scaling'] = self.error_scaling
        dump['scalings'] = self.scalings
        return dump

    def optimize(self, loaded_checkpoint=None):
        """variational_inference.py:340-394."""
        if loaded_checkpoint is None:
            params = self._initialize()
        else:
            params = [loaded_checkpoint[name] for name in self.param_names]
            try:
                self.error_scaling = loaded_checkpoint['error_scaling']
            except KeyError:
                logging.warning('Did not find "error_scaling" in the loaded checkpoint.')
            self._set_state(params)
        converged = False
        elbo = self.elbo(params)
        running = None
        num_its = 0
        L = np.ones(5)
        post_mean = self.real_posterior_mean(*params)
        while num_its < self.num_its and not converged:
            if self.checkpoint and num_its % self.checkpoint_freq == 0:
                np.savez('{}.{}'.format(self.checkpoint_path, num_its),
                         **self.create_dump_dict(params))
            new_params, L, elbo, running = self._optimize_step(
                params, L=L, curr_elbo=elbo, line_search_rate=2., running_elbo_delta=running)
            new_post_mean = self.real_posterior_mean(*new_params)
            converged = np.allclose(new_post_mean, post_mean, atol=ABS_TOL, rtol=REL_TOL)
            converged = converged or np.isclose(running, 0, atol=ELBO_TOL, rtol=0)
            if num_its < 10 and loaded_checkpoint is None:
                converged = False
            post_mean = new_post_mean
            num_its += 1
            params = tuple(new_params)
        if num_its == self.num_its:
            logging.warning('Failed to converge')
        logging.info('Optimization ran for %d iterations', num_its)
        self.num_its_run = num_its
        return params
