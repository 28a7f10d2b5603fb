#!/usr/bin/env python3
"""Generate golden vectors for the `vilma fit` hot path from the REFERENCE itself.

Runs ONLY in the build container (the reference lives at /root/reference and never
travels).  The reference is imported by path with three inert stand-ins for packages
that are not installed here (numba -> identity decorators, h5py/plinkio -> empty
modules; see tests/golden/_shim).  The jitted bodies of vilma.numerics are ordinary
Python, so the semantics are the reference's own up to summation order.

    PYTHONPATH=tests/golden/_shim:/root/reference/src PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/make_golden.py

Outputs (committed, data only):
    tests/golden/numerics_kat.npz          per-function known-answer vectors (numerics.py)
    tests/golden/ldop_kat.npz              LowRankMatrix / BlockDiagonalMatrix vectors
    tests/golden/vischeme_kat.npz          MultiPopVI behaviours on the reference tests' own problems
    tests/golden/traj_<name>.npz           class-API trajectories driven through
                                           MultiPopVI._optimize_step (variational_inference.py:396)
    tests/golden/mixgrid_kat.npz           vi_options._make_simple outputs (RNG order pin)
    tests/golden/loader_kat.npz            load.py outputs on the reference's own fixtures
    tests/golden/sim_kat.npz               sim.py draws for pinned seeds (RNG order pin of `vilma sim`)
"""
import io
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

from vilma import numerics as rn                      # noqa: E402
from vilma import matrix_structures as rms            # noqa: E402
from vilma import variational_inference as rvi        # noqa: E402
from vilma import vi_options as rvo                   # noqa: E402
from vilma import load as rload                       # noqa: E402
from vilma import sim as rsim                         # noqa: E402


# --------------------------------------------------------------------------------------
# synthetic inputs (recipe of SURVEY.md section 8d, scaled down)
# --------------------------------------------------------------------------------------
def ar1(n, rho):
    idx = np.arange(n)
    return rho ** np.abs(idx[:, None] - idx[None, :])


def factor_ld(rng, n):
    m = max(2, int(np.ceil(n / 4)))
    f = rng.normal(size=(n, m))
    cov = f @ f.T / m + 0.05 * np.eye(n)
    d = 1.0 / np.sqrt(np.diag(cov))
    return cov * np.outer(d, d)


def make_problem(seed, P, sizes, M, ldthresh, kind, frac_missing, A, shuffle):
    rng = np.random.default_rng(seed)
    n_ld = int(np.sum(sizes))
    n_missing = int(round(frac_missing * n_ld))
    N = n_ld + n_missing
    # which SNP (extract order) sits at each LD-order position
    order = rng.permutation(N) if shuffle else np.arange(N)
    ld_members = order[:n_ld]
    if not shuffle:
        ld_members = np.sort(rng.choice(N, size=n_ld, replace=False))
    missing = np.setdiff1d(np.arange(N), ld_members)
    blocks = []     # blocks[p][b] = dense symmetric matrix handed to LowRankMatrix(X, t)
    rhos = np.zeros((P, len(sizes)))
    for p in range(P):
        this = []
        for b, n in enumerate(sizes):
            if kind == 'ar1':
                rhos[p, b] = rng.uniform(0.5, 0.95)
                this.append(ar1(n, rhos[p, b]))
            else:
                this.append(factor_ld(rng, n))
        blocks.append(this)
    perm = np.concatenate([ld_members, missing]).astype(np.int64)

    se = rng.uniform(0.005, 0.02, size=(P, N))
    causal = rng.random(N) < 0.05
    cov = 0.01 ** 2 * (0.2 * np.eye(P) + 0.8 * np.ones((P, P)))
    beta = np.zeros((P, N))
    beta[:, causal] = rng.multivariate_normal(np.zeros(P), cov, size=int(causal.sum())).T
    betahat = np.zeros((P, N))
    for p in range(P):
        start = 0
        z_true = (beta[p] / se[p])[perm]
        out = np.zeros(N)
        for X in blocks[p]:
            n = X.shape[0]
            w, v = np.linalg.eigh(X)
            w = np.maximum(w, 0)
            half = (v * np.sqrt(w)) @ v.T
            out[start:start + n] = X @ z_true[start:start + n] + half @ rng.normal(size=n)
            start += n
        inv_perm = np.argsort(perm)
        betahat[p] = se[p] * out[inv_perm]
    betahat[:, missing] = 0.0        # the loader sets BETA=0, SE=1 for SNPs it cannot use
    se[:, missing] = 1.0

    var = np.geomspace(1e-8, 1e-2, M)
    covs = []
    for k in range(M):
        if P == 1:
            covs.append(np.array([[var[k]]]))
        else:
            r = (0.0, 0.5, 0.9)[k % 3]
            covs.append(var[k] * ((1 - r) * np.eye(P) + r * np.ones((P, P))))
    annotations = np.zeros((N, A))
    annotations[np.arange(N), rng.integers(0, A, size=N)] = 1
    return dict(P=P, N=N, M=M, A=A, sizes=np.asarray(sizes, dtype=np.int64), perm=perm,
                missing=missing.astype(np.int64), blocks=blocks, ldthresh=ldthresh,
                betahat=betahat, se=se, covs=np.array(covs), annotations=annotations,
                kind=kind, rhos=rhos)


def build_reference_vi(prob, scaled, scale_se, num_its, gwas_N, init_hg):
    ld_mats = []
    for p in range(prob['P']):
        mats = [rms.LowRankMatrix(X, prob['ldthresh']) for X in prob['blocks'][p]]
        ld_mats.append(rms.BlockDiagonalMatrix(mats, perm=prob['perm'],
                                               missing=prob['missing']))
    vi = rvi.MultiPopVI(marginal_effects=prob['betahat'], std_errs=prob['se'],
                        ld_mats=ld_mats, mixture_covs=list(prob['covs']),
                        annotations=prob['annotations'], checkpoint=False,
                        checkpoint_freq=-1, output='golden', scaled=scaled,
                        scale_se=scale_se, gwas_N=np.asarray(gwas_N, dtype=float),
                        init_hg=np.asarray(init_hg, dtype=float), num_its=num_its)
    return vi, ld_mats


class ObjectiveLog(logging.Handler):
    """Collects the sequence of line-search trial objectives the reference logs."""
    def __init__(self):
        super().__init__(level=logging.INFO)
        self.records = []

    def emit(self, record):
        self.records.append((record.msg, record.args))


def trajectory(name, prob, n_sweeps, scaled=False, scale_se=False, seed=42,
               gwas_N=None, init_hg=None, full_optimize_its=None, compact=False):
    """compact=True (mid-size problems): AR(1) LD is stored as its rho per (cohort, block) --
    tests rebuild R_ij = rho^|i-j| -- and the [M,P,P,N] / initial arrays are left out."""
    P = prob['P']
    gwas_N = [1e5] * P if gwas_N is None else gwas_N
    init_hg = [0.1] * P if init_hg is None else init_hg
    vi, ld_mats = build_reference_vi(prob, scaled, scale_se, n_sweeps, gwas_N, init_hg)

    # instrument: count LD matvecs and record every objective the loop evaluates
    counts = {'dot': 0}
    for ld in ld_mats:
        orig = ld.dot

        def counted(vec, _orig=orig):
            counts['dot'] += 1
            return _orig(vec)
        ld.dot = counted
    objs = []
    orig_elbo, orig_bobj = vi.elbo, vi._beta_objective

    def elbo_logged(params):
        v = orig_elbo(params)
        objs.append((0.0, float(v)))
        return v

    def bobj_logged(params):
        v = orig_bobj(params)
        objs.append((1.0, float(v)))
        return v
    vi.elbo, vi._beta_objective = elbo_logged, bobj_logged

    np.random.seed(seed)
    params = vi._initialize()
    out = {}
    out['init_vi_mu'], out['init_vi_delta'], out['init_hyper_delta'] = [np.copy(x) for x in params]
    out['init_nat_grad_vi_delta'] = np.copy(vi.nat_grad_vi_delta)
    elbo = vi.elbo(params)
    out['init_elbo'] = elbo
    L = np.ones(5)
    red = None
    elbos, Ls, reds, taus, hypers, pmeans, dots, nobj = [], [], [], [], [], [], [], []
    for it in range(n_sweeps):
        d0, o0 = counts['dot'], len(objs)
        params, L, elbo, red = vi._optimize_step(params, L=L, curr_elbo=elbo,
                                                 line_search_rate=2.,
                                                 running_elbo_delta=red)
        params = tuple(params)
        elbos.append(float(elbo)); Ls.append(np.copy(L)); reds.append(float(red))
        taus.append(np.copy(vi.error_scaling)); hypers.append(np.copy(params[2]))
        pmeans.append(vi.real_posterior_mean(*params))
        dots.append(counts['dot'] - d0); nobj.append(len(objs) - o0)
        print('  %s sweep %d elbo %.10f L0 %.4f dots %d' % (name, it, elbo, L[0], dots[-1]),
              flush=True)
    out.update(
        P=P, N=prob['N'], M=prob['M'], A=prob['A'], sizes=prob['sizes'], perm=prob['perm'],
        missing=prob['missing'], ldthresh=prob['ldthresh'], betahat=prob['betahat'],
        se=prob['se'], covs=prob['covs'], annotations=prob['annotations'],
        scaled=scaled, scale_se=scale_se, seed=seed, gwas_N=np.asarray(gwas_N, dtype=float),
        init_hg=np.asarray(init_hg, dtype=float),
        # derived constants (VIScheme.__init__, variational_inference.py:189-252)
        ld_diags=vi.ld_diags, adj_marginal_effects=vi.adj_marginal_effects,
        chi_stat=vi.chi_stat, ld_ranks=vi.ld_ranks, inverse_betas=vi.inverse_betas,
        scalings=vi.scalings, mixture_prec=vi.mixture_prec, log_det=vi.log_det,
        # trajectory
        elbo=np.array(elbos), L=np.array(Ls), running_elbo_delta=np.array(reds),
        error_scaling=np.array(taus), hyper_delta=np.array(hypers),
        post_mean=np.array(pmeans), dots_per_sweep=np.array(dots),
        objs_per_sweep=np.array(nobj), objective_log=np.array(objs),
        final_vi_mu=params[0], final_vi_delta=params[1], final_hyper_delta=params[2],
        final_post_var=vi.real_posterior_variance(*params), final_vi_sigma=vi.vi_sigma,
    )
    if compact:
        assert prob['kind'] == 'ar1'
        out['ld_rho'] = prob['rhos']
        for key in ('init_vi_mu', 'init_vi_delta', 'init_nat_grad_vi_delta', 'final_vi_sigma',
                    'final_vi_delta'):
            out.pop(key)
        out['post_mean'] = out['post_mean'][[0, len(elbos) // 2, len(elbos) - 1]]
        out['post_mean_sweeps'] = np.array([0, len(elbos) // 2, len(elbos) - 1])
    for p in range(P):
        if not compact:
            for b, X in enumerate(prob['blocks'][p]):
                out['ld_%d_%d' % (p, b)] = X
        for b, m in enumerate(ld_mats[p].matrices):
            out['rank_%d_%d' % (p, b)] = m.s.shape[0]

    if full_optimize_its is not None:
        # a second, independent object driven through optimize() to pin the convergence
        # rule and iteration count (variational_inference.py:340-394)
        vi2, _ = build_reference_vi(prob, scaled, scale_se, full_optimize_its, gwas_N, init_hg)
        handler = ObjectiveLog()
        root = logging.getLogger()
        old_level = root.level
        root.addHandler(handler); root.setLevel(logging.INFO)
        np.random.seed(seed)
        fin = vi2.optimize()
        root.removeHandler(handler); root.setLevel(old_level)
        n_it = [a[0] for m, a in handler.records if m.startswith('Optimization ran for')]
        out['opt_num_its'] = int(n_it[0])
        out['opt_vi_mu'], out['opt_vi_delta'], out['opt_hyper_delta'] = fin
        out['opt_error_scaling'] = vi2.error_scaling
        out['opt_post_mean'] = vi2.real_posterior_mean(*fin)
        print('  %s optimize(): %d iterations' % (name, out['opt_num_its']), flush=True)
    np.savez_compressed(os.path.join(HERE, 'traj_%s.npz' % name), **out)


# --------------------------------------------------------------------------------------
# per-function KATs (numerics.py, one recipe per function as in tests/test.py:877-1217)
# --------------------------------------------------------------------------------------
def numerics_kat():
    rng = np.random.default_rng(7)
    out = {}
    for P, M, N, A in ((1, 5, 37, 1), (2, 4, 50, 2), (3, 6, 21, 3)):
        t = 'P%d_' % P
        mu = rng.normal(size=(M, P, N))
        mu2 = rng.normal(size=(M, P, N))
        delta = rng.dirichlet(np.ones(M), size=N)
        B = rng.normal(size=(M, P, P, N))
        lam = np.einsum('kpqi,krqi->kpri', B, B) + 0.5 * np.eye(P)[None, :, :, None]
        prec = lam[:, :, :, :1].copy()
        ann = rng.integers(0, A, size=N).astype(np.int64)
        hyper = rng.dirichlet(np.ones(M), size=A)
        log_det = rng.normal(size=M)
        x = rng.normal(size=(P, N)); y = rng.uniform(0.5, 2, size=(P, N))
        w = rng.normal(size=(P, N)); z = rng.normal(size=(P, N))
        chi = rng.uniform(1, 2, size=P); ranks = rng.integers(5, 10, size=P).astype(float)
        tau = rng.uniform(0.5, 2, size=P)
        const = rng.normal(size=(N, M)); natd = rng.normal(size=(N, M - 1))
        out.update({
            t + 'mu': mu, t + 'mu2': mu2, t + 'delta': delta, t + 'lam': lam, t + 'prec': prec,
            t + 'ann': ann, t + 'hyper': hyper, t + 'log_det': log_det, t + 'x': x, t + 'y': y,
            t + 'w': w, t + 'z': z, t + 'chi': chi, t + 'ranks': ranks, t + 'tau': tau,
            t + 'const': const, t + 'natd': natd,
            t + 'sum_betas': rn.sum_betas(mu, mu2, 0.3),
            t + 'fast_divide': rn.fast_divide(x, y),
            t + 'fast_linked_ests': rn.fast_linked_ests(w, y, x, z),
            t + 'fast_likelihood': rn.fast_likelihood(x, y, w, z, mu[0], mu2[0], chi, ranks, tau),
            t + 'fast_posterior_mean': rn.fast_posterior_mean(mu, delta),
            t + 'fast_pmv': rn.fast_pmv(rn.fast_posterior_mean(mu, delta), mu, delta,
                                        np.abs(mu2)),
            t + 'fast_nat_inner_product_m2': rn.fast_nat_inner_product_m2(mu, lam),
            t + 'fast_nat_inner_product': rn.fast_nat_inner_product(mu, lam),
            t + 'fast_inner_product_comp': rn.fast_inner_product_comp(mu, prec, delta),
            t + 'sum_annotations': rn.sum_annotations(delta, ann, A),
            t + 'fast_delta_kl': rn.fast_delta_kl(delta, hyper, ann),
            t + 'fast_beta_kl': rn.fast_beta_kl(const, delta),
            t + 'fast_vi_delta_grad': rn.fast_vi_delta_grad(hyper, log_det, ann),
            t + 'map_to_nat_cat_2D': rn.map_to_nat_cat_2D(delta),
            t + 'invert_nat_cat_2D': rn.invert_nat_cat_2D(natd * 40),
            t + 'fast_invert_nat_vi_delta': rn.fast_invert_nat_vi_delta(mu, mu2, const, natd),
            t + 'vi_sigma_inv': rn.vi_sigma_inv(lam),
            t + 'vi_sigma_log_det': rn.vi_sigma_log_det(lam),
        })
    np.savez_compressed(os.path.join(HERE, 'numerics_kat.npz'), **out)


def ldop_kat():
    """LowRankMatrix / BlockDiagonalMatrix vectors incl. a rank-deficient block, the
    degenerate all-below-threshold block and perm/missing (matrix_structures.py)."""
    rng = np.random.default_rng(11)
    out = {}
    sizes = [7, 12, 5]
    mats = []
    for b, n in enumerate(sizes):
        X = factor_ld(rng, n)
        if b == 1:                       # make it singular: duplicate a SNP
            X[:, 3] = X[:, 2]; X[3, :] = X[2, :]; X[3, 3] = 1.0
        mats.append(X)
        out['X%d' % b] = X
    N = sum(sizes) + 3
    perm = rng.permutation(N).astype(np.int64)
    missing = perm[-3:].copy()
    out['perm'] = perm; out['missing'] = missing
    vec = rng.normal(size=N); out['vec'] = vec
    reg = rng.uniform(0.5, 2.0, size=N); out['reg'] = reg
    for t in (1.0, 0.8, 0.3):
        lrs = [rms.LowRankMatrix(X, t) for X in mats]
        bd = rms.BlockDiagonalMatrix(lrs, perm=perm, missing=missing)
        tag = 't%02d_' % int(t * 10)
        out[tag + 'dot'] = bd.dot(vec)
        out[tag + 'inv_dot'] = bd.inverse.dot(vec)
        out[tag + 'ridge'] = bd.ridge_inverse_dot(vec, reg)
        out[tag + 'ridge_scalar'] = bd.ridge_inverse_dot(vec, 0.7)
        out[tag + 'diag'] = bd.diag()
        out[tag + 'dot_i'] = np.array([bd.dot_i(vec, i) for i in range(N)])
        out[tag + 'sqrt_dot'] = bd.matrix_power(0.5).dot(vec)
        out[tag + 'inv_inv_dot'] = bd.inverse.inverse.dot(vec)
        out[tag + 'dot_matrix'] = bd.dot(np.stack([vec, 2 * vec - 1], axis=1))
        out[tag + 'rank'] = bd.get_rank()
        out[tag + 'ranks'] = np.array([m.get_rank() for m in lrs])
        out[tag + 'starts'] = bd.starts
        out[tag + 'inv_perm'] = bd.inv_perm
        for b, m in enumerate(lrs):
            out[tag + 's%d' % b] = m.s
            out[tag + 'recon%d' % b] = (np.asarray(m.u) * m.s) @ np.asarray(m.v)
    # degenerate block: every eigenvalue below the threshold -> rank-1 zero operator
    Xs = 0.01 * np.eye(4)
    lr = rms.LowRankMatrix(Xs, 0.5)
    out['degenerate_X'] = Xs
    out['degenerate_s'] = lr.s
    out['degenerate_rank'] = lr.get_rank()
    out['degenerate_dot'] = lr.dot(np.arange(4.0))
    np.savez_compressed(os.path.join(HERE, 'ldop_kat.npz'), **out)


def vischeme_problem(linked, num_annotations):
    """The two little problems the reference's own behavioural tests are written on
    (tests/test.py:1225-1293): 2 cohorts x 50 SNPs, one LD block (a fixed dense correlation
    matrix, or the identity for the "unlinked" variant), two mixture components."""
    if linked:
        betas = np.arange(100).reshape(2, 50).astype(float)
        ld = (1 + np.arange(50 * 50)).reshape(50, 50) / (50 * 50 + 1)
        ld = ld + ld.T + 5 * np.eye(50)
        d = np.diag(1 / np.sqrt(np.diag(ld)))
        ld = d @ ld @ d
    else:
        betas = np.arange(100).reshape(50, 2).T.astype(float)
        ld = np.eye(50)
    std_errs = np.array([1.] * 50 + [2.] * 50).reshape(2, 50)
    if num_annotations == 2:
        ann = np.zeros((50, 2), dtype=int)
        ann[0:25, 0] = 1
        ann[25:, 1] = 1
    else:
        ann = np.ones((50, 1), dtype=int)
    return dict(betas=betas, std_errs=std_errs, ld=ld, annotations=ann,
                mixture_covs=[np.eye(2), 2 * np.eye(2)], gwas_N=np.array([100e3, 10e3]),
                init_hg=np.array([0.1, 0.9]))


def vischeme_kat():
    """What the reference's MultiPopVI computes on those problems: construction-time constants,
    _initialize, _update_error_scaling, _nat_to_not_vi_delta, _update_beta (twice: idempotence),
    _nat_grad_step, a few sweeps and optimize() -- the behaviours tests/test.py:1297-1411,
    1473-1514, 1582-1604, 1704-1726, 1849-1876 assert."""
    out = {}
    for tag, linked, A, scaled, scale_se in (('linked_a2', True, 2, False, False),
                                             ('unlinked_a1', False, 1, False, False),
                                             ('linked_a1_scaled', True, 1, True, False),
                                             ('linked_a2_scale_se', True, 2, False, True),
                                             ('linked_a2_scaled_scale_se', True, 2, True, True)):
        pr = vischeme_problem(linked, A)
        lr = rms.LowRankMatrix(X=pr['ld'], t=1.0)
        vi = rvi.MultiPopVI(marginal_effects=pr['betas'], std_errs=pr['std_errs'],
                            ld_mats=[rms.BlockDiagonalMatrix([lr]), rms.BlockDiagonalMatrix([lr])],
                            mixture_covs=pr['mixture_covs'], annotations=pr['annotations'],
                            checkpoint=False, checkpoint_freq=-1, output='test', scaled=scaled,
                            scale_se=scale_se, gwas_N=pr['gwas_N'], init_hg=pr['init_hg'],
                            num_its=20)
        t = tag + '_'
        for key in ('vi_sigma', 'nat_sigma', 'vi_sigma_log_det', 'vi_sigma_matches',
                    'sigma_summary', 'adj_marginal_effects', 'chi_stat', 'ld_ranks',
                    'inverse_betas', 'scaled_ld_diags', 'ld_diags', 'scalings', 'log_det'):
            out[t + key] = np.array(getattr(vi, key))
        np.random.seed(42)
        mu, delta, hyper = vi._initialize()
        params = (mu, delta, hyper)
        out[t + 'init_mu'], out[t + 'init_delta'], out[t + 'init_hyper'] = mu, delta, hyper
        out[t + 'init_elbo'] = vi.elbo(params)
        out[t + 'init_post_mean'] = vi.real_posterior_mean(*params)
        out[t + 'init_post_var'] = vi.real_posterior_variance(*params)
        # _nat_to_not_vi_delta at the initial point
        out[t + 'fixed_point_delta'] = vi._nat_to_not_vi_delta(params)[1]
        # _update_error_scaling at the initial point, then back to tau = 1
        vi._update_error_scaling(params)
        out[t + 'tau_after_update'] = np.array(vi.error_scaling)
        out[t + 'elbo_after_tau'] = vi.elbo(vi._nat_to_not_vi_delta(params))
        vi.error_scaling = np.ones(2)
        vi._set_vi_sigma()
        # _update_beta twice from the fixed point of the initial state
        fp = vi._nat_to_not_vi_delta(params)
        p1, L1, o1, n1 = vi._update_beta(*fp, None, [1., 1., 1.], 0, 1.25)
        p2, L2, o2, n2 = vi._update_beta(*p1, None, [1., 1., 1.], 0, 1.25)
        out[t + 'ub1_mu'], out[t + 'ub1_delta'] = p1[0], p1[1]
        out[t + 'ub1_objs'] = np.array([o1, n1]); out[t + 'ub1_L'] = np.array(L1)
        out[t + 'ub2_mu'], out[t + 'ub2_delta'] = p2[0], p2[1]
        out[t + 'ub2_objs'] = np.array([o2, n2]); out[t + 'ub2_L'] = np.array(L2)
        # _update_hyper_delta from there
        p3, L3, o3, n3 = vi._update_hyper_delta(*p1, None, [1., 1., 1.], 1, 1.25)
        out[t + 'uh_hyper'], out[t + 'uh_delta'] = p3[2], p3[1]
        out[t + 'uh_objs'] = np.array([o3, n3])
        # _nat_grad_step from the fixed point, from L = 1 and from close to L_MAX
        vi.error_scaling = np.ones(2)
        vi._set_vi_sigma()
        vi.nat_grad_vi_delta = rn.fast_vi_delta_grad(hyper, vi.log_det, vi.annotations)
        q, Lq, dq = vi._nat_grad_step(fp, [1., 1., 1.], 2., None)
        out[t + 'ngs_mu'], out[t + 'ngs_hyper'] = q[0], q[2]
        out[t + 'ngs_L'], out[t + 'ngs_delta_elbo'] = np.array(Lq), dq
        out[t + 'ngs_tau'] = np.array(vi.error_scaling)
        # optimize() from the same seed
        vi.error_scaling = np.ones(2)
        vi._set_vi_sigma()
        np.random.seed(42)
        final = vi.optimize()
        out[t + 'opt_elbo'] = vi.elbo(final)
        out[t + 'opt_post_mean'] = vi.real_posterior_mean(*final)
        out[t + 'opt_tau'] = np.array(vi.error_scaling)
        out[t + 'opt_hyper'] = final[2]
    np.savez_compressed(os.path.join(HERE, 'vischeme_kat.npz'), **out)


def mixgrid_kat():
    out = {}
    for P, K in ((1, 5), (2, 3), (3, 2)):
        np.random.seed(42)
        mins = np.linspace(1e-6, 2e-6, P); maxes = np.linspace(1e-2, 3e-2, P)
        covs = rvo._make_simple(P, K, mins, maxes)
        out['P%d_K%d' % (P, K)] = np.array(covs)
        out['P%d_K%d_mins' % (P, K)] = mins
        out['P%d_K%d_maxes' % (P, K)] = maxes
        out['P%d_K%d_next_uniform' % (P, K)] = np.random.uniform()
    np.savez_compressed(os.path.join(HERE, 'mixgrid_kat.npz'), **out)


def loader_kat():
    """Reference loaders on the reference's own fixtures (copied to tests/golden/refdata and
    tests/golden/example): pins perm / missing / block contents / sumstats alignment."""
    import warnings
    warnings.simplefilter('ignore')
    ref = os.path.join(HERE, 'refdata')
    ex = os.path.join(HERE, 'example')
    out = {}
    cases = {
        'plain': ('ld_manifest.tsv', 'good_variants.tsv', [], 1.0),
        'thresh': ('ld_manifest.tsv', 'good_variants.tsv', [], 0.8),
        'deny': ('ld_manifest.tsv', 'good_variants.tsv', [3, 4, 5], 1.0),
        'svd': ('ld_manifest_svd.tsv', 'good_variants.tsv', [], 1.0),
        'svd_deny': ('ld_manifest_svd.tsv', 'good_variants.tsv', [3, 4, 5], 0.8),
        'plusmissing': ('ld_manifest.tsv', 'good_variants_plus_missing.tsv', [], 1.0),
    }
    for tag, (manifest, varfile, deny, t) in cases.items():
        variants = rload.load_variant_list(os.path.join(ref, varfile))
        bd, miss = rload.load_ld_from_schema(os.path.join(ref, manifest), variants, deny, t, False)
        out[tag + '_perm'] = bd.perm
        out[tag + '_missing_list'] = np.array(miss, dtype=np.int64)
        out[tag + '_missing'] = bd.missing
        out[tag + '_starts'] = bd.starts
        out[tag + '_diag'] = bd.diag()
        out[tag + '_rank'] = bd.get_rank()
        for b, m in enumerate(bd.matrices):
            out[tag + '_recon%d' % b] = (np.asarray(m.u) * m.s) @ np.asarray(m.v)
            out[tag + '_s%d' % b] = m.s
        v = np.linspace(-1, 1, bd.shape[0])
        out[tag + '_dot'] = bd.dot(v)
        out[tag + '_invdot'] = bd.inverse.dot(v)
    variants = rload.load_variant_list(os.path.join(ex, 'keep_variants.txt'))
    bd, miss = rload.load_ld_from_schema(os.path.join(ex, 'ld_mat', 'example_schema.schema'),
                                         variants, [], 1.0, False)
    out['example_perm'] = bd.perm
    out['example_missing_list'] = np.array(miss, dtype=np.int64)
    out['example_starts'] = bd.starts
    for b, m in enumerate(bd.matrices):
        out['example_recon%d' % b] = (np.asarray(m.u) * m.s) @ np.asarray(m.v)
    variants = rload.load_variant_list(os.path.join(ref, 'good_variants.tsv'))
    for name in ('good_sumstats_beta', 'good_sumstats_or', 'good_sumstats_flip'):
        st, miss = rload.load_sumstats(os.path.join(ref, name + '.tsv'), variants)
        out[name + '_BETA'] = np.array(st.BETA, dtype=float)
        out[name + '_SE'] = np.array(st.SE, dtype=float)
        out[name + '_missing'] = np.array(miss, dtype=np.int64)
    vpm = rload.load_variant_list(os.path.join(ref, 'good_variants_plus_missing.tsv'))
    st, miss = rload.load_sumstats(os.path.join(ref, 'good_sumstats_beta_plus_missing.tsv'), vpm)
    out['plusmissing_BETA'] = np.array(st.BETA, dtype=float)
    out['plusmissing_SE'] = np.array(st.SE, dtype=float)
    out['plusmissing_sumstats_missing'] = np.array(miss, dtype=np.int64)
    ann, deny = rload.load_annotations(os.path.join(ref, 'good_annotations.tsv'), variants)
    out['annotations'] = np.asarray(ann, dtype=float)
    out['annotations_denylist'] = np.array(deny, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'loader_kat.npz'), **out)


def sim_kat():
    """`vilma sim` building blocks with the legacy generator seeded: pins the order in which
    the random stream is consumed (component per SNP, then effects, then sampling noise)."""
    out = {}
    rng = np.random.default_rng(11)
    N, A, M, P = 300, 3, 5, 2
    ann = np.zeros((N, A))
    ann[np.arange(N), rng.integers(0, A, size=N)] = 1
    w = rng.dirichlet(np.ones(M), size=A)
    base = rng.normal(size=(M, P, P))
    covs = np.einsum('kij,klj->kil', base, base) + 0.1 * np.eye(P)
    out['annotations'], out['weights'], out['covs'] = ann, w, covs
    np.random.seed(7)
    out['components'] = rsim.sim_components(ann, w)
    out['after_components_uniform'] = np.random.uniform()
    np.random.seed(8)
    out['true_effects'] = rsim.sim_true_effects(ann, w, covs)
    out['after_effects_normal'] = np.random.normal()
    # sim_gwas on a small block-diagonal operator with missing SNPs and a shuffled perm
    sizes = [7, 5, 9]
    blocks = [rms.LowRankMatrix(ar1(n, 0.6 + 0.1 * b), 0.999999) for b, n in enumerate(sizes)]
    n_ld = sum(sizes)
    total = n_ld + 3
    perm = rng.permutation(total)
    bd = rms.BlockDiagonalMatrix(blocks, perm=perm, missing=perm[n_ld:])
    beta = rng.normal(size=total) * 1e-2
    se = rng.uniform(0.01, 0.05, size=total)
    np.random.seed(9)
    out['gwas_blocks'] = np.array(sizes)
    out['gwas_rho'] = np.array([0.6 + 0.1 * b for b in range(len(sizes))])
    out['gwas_perm'], out['gwas_beta'], out['gwas_se'] = perm, beta, se
    out['gwas_betahat'] = rsim.sim_gwas(beta, se, bd)
    # the whole command on the reference's own fixtures (its test_cli_sim, tests/test.py:
    # 2200-2245).  The .tsv shipped with the reference as the expected output of that test is
    # NOT reproduced by the reference's own code under the numpy/pandas of this container, so
    # the expected output is regenerated here; --mmap is forced off (h5py is not installed).
    import argparse
    import tempfile
    import warnings
    import pandas as pd
    warnings.simplefilter('ignore')
    ref = os.path.join(HERE, 'refdata')
    load_ld = rload.load_ld_from_schema
    rload.load_ld_from_schema = lambda *a, **k: load_ld(*a, **dict(k, mmap=False))
    try:
        with tempfile.TemporaryDirectory() as tmp:
            for tag in ('npy', 'npz'):
                prefix = os.path.join(tmp, 'run_' + tag)
                rsim.main(argparse.Namespace(
                    ld_schema=os.path.join(ref, 'ld_manifest.tsv'),
                    sumstats=os.path.join(ref, 'good_sumstats_beta.tsv'),
                    annotations=os.path.join(ref, 'good_annotations.tsv'),
                    covariance=os.path.join(ref, 'copy_vilma_run.covariance.pkl'),
                    weights=os.path.join(ref, 'sim_weights.' + tag), output=prefix,
                    names='simpop1', seed=143, gwas_n_scaling='1.'))
                table = pd.read_csv(prefix + '.simpop1.simgwas.tsv', sep='\t')
                out['cli_%s_columns' % tag] = np.array(list(table.columns))
                out['cli_%s_ID' % tag] = np.array(table.ID, dtype=str)
                for col in ('SE', 'BETA', 'true_beta'):
                    out['cli_%s_%s' % (tag, col)] = np.array(table[col], dtype=float)
    finally:
        rload.load_ld_from_schema = load_ld
    np.savez_compressed(os.path.join(HERE, 'sim_kat.npz'), **out)


def main():
    which = set(sys.argv[1:])

    def want(name):
        return not which or name in which

    if want('kat'):
        numerics_kat(); ldop_kat(); mixgrid_kat(); loader_kat()
    if want('kat') or want('vischeme'):
        vischeme_kat()
    if want('ldop'):
        ldop_kat()
    if want('kat') or want('sim'):
        sim_kat()
    if want('p1_dense'):
        prob = make_problem(1, P=1, sizes=[90, 60, 120, 75, 100, 55], M=25, ldthresh=1.0,
                            kind='ar1', frac_missing=0.0, A=1, shuffle=False)
        trajectory('p1_dense', prob, n_sweeps=12, full_optimize_its=40)
    if want('p2_lowrank'):
        prob = make_problem(2, P=2, sizes=[80, 110, 60, 95, 70], M=40, ldthresh=0.8,
                            kind='factor', frac_missing=0.05, A=1, shuffle=True)
        trajectory('p2_lowrank', prob, n_sweeps=10)
    if want('p2_scale_se'):
        prob = make_problem(3, P=2, sizes=[70, 90, 50, 85], M=12, ldthresh=1.0,
                            kind='ar1', frac_missing=0.05, A=2, shuffle=True)
        trajectory('p2_scale_se', prob, n_sweeps=14, scale_se=True, gwas_N=[1e5, 5e4],
                   init_hg=[0.1, 0.3], full_optimize_its=30)
    if want('p4_general'):
        prob = make_problem(4, P=4, sizes=[60, 45, 70], M=12, ldthresh=0.9,
                            kind='factor', frac_missing=0.04, A=1, shuffle=False)
        trajectory('p4_general', prob, n_sweeps=8)
    if want('p1_scaled'):
        prob = make_problem(5, P=1, sizes=[64, 80, 50], M=10, ldthresh=1.0,
                            kind='ar1', frac_missing=0.03, A=3, shuffle=True)
        trajectory('p1_scaled', prob, n_sweeps=10, scaled=True, scale_se=True)
    if want('p4_m81'):
        # the shape of BASELINE.json configs[4]: 4 cohorts, M = 81 (the P > 2 branch of
        # numerics.py:238-290 with a large component count), blocks wider than one 128-column slab
        prob = make_problem(6, P=4, sizes=[140, 135], M=81, ldthresh=1.0,
                            kind='ar1', frac_missing=0.03, A=1, shuffle=True)
        trajectory('p4_m81', prob, n_sweeps=6)
    if want('p2_bigblock'):
        # a 330-SNP block: three slabs of the symmetric dense product, two combine chunks
        prob = make_problem(7, P=2, sizes=[330, 140], M=10, ldthresh=1.0,
                            kind='ar1', frac_missing=0.02, A=1, shuffle=True)
        trajectory('p2_bigblock', prob, n_sweeps=8)
    if want('p2_bigblock_lr'):
        # the same shape with eigen-truncated factor-model LD (--ldthresh 0.8): multi-slab U
        prob = make_problem(8, P=2, sizes=[330, 140], M=10, ldthresh=0.8,
                            kind='factor', frac_missing=0.02, A=1, shuffle=True)
        trajectory('p2_bigblock_lr', prob, n_sweeps=8)
    if want('p2_mid'):
        # mid-size (5 000 LD SNPs + 3 % missing, 2 cohorts): long enough a run to hold sweeps with
        # several beta updates, rejected first steps and a both-candidates-rejected trial; LD stored
        # as AR(1) parameters (compact)
        prob = make_problem(9, P=2, sizes=[600, 450, 520, 380, 700, 350, 500, 480, 420, 600],
                            M=20, ldthresh=1.0, kind='ar1', frac_missing=0.03, A=1, shuffle=True)
        trajectory('p2_mid', prob, n_sweeps=16, gwas_N=[2e5, 2e5], init_hg=[0.5, 0.5],
                   compact=True)


if __name__ == '__main__':
    main()
