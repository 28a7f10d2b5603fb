"""Synthetic fit problems of the BASELINE.json shapes (recipe of SURVEY.md section 8d).

Everything random is drawn from numpy Generators seeded per (seed, cohort, block), so the
global problem is identical however the blocks are sharded over GPUs -- the 1/2/4/8-GPU runs
of bench.py fit the same data.  LD is AR(1) per (block, cohort), which is positive definite:
the reference's eigen-decomposition at ldthresh=1 keeps every eigenpair, so the operator is
the matrix itself, R^+ = R^-1 and rank = n -- the load-time constants of VIScheme.__init__
(reference variational_inference.py:236-252) are computed here in closed form: the Cholesky
factor of an AR(1) matrix is the AR(1) recursion and its inverse is tridiagonal, so sumstats,
chi statistics and the ridge start cost O(n) per block on the host (no per-block
factorisation of the 3 400 / 34 000 block-cohorts of workloads C3 / C5).

SNP layout: block after block, each block's LD SNPs followed by the LD-missing SNPs attached
to it (`missing_frac` of all SNPs, multinomially spread), so `perm` is not the identity.
"""
import sys

import numpy as np

WORKLOADS = {
    # name: (P, N_ld, B, M, fixed block size or None)
    'C2': dict(P=1, n_ld=100_000, B=500, M=25, fixed=200, missing_frac=0.0),
    'C3': dict(P=2, n_ld=1_000_000, B=1700, M=40, fixed=None, missing_frac=0.05),
    'tiny': dict(P=2, n_ld=6_000, B=12, M=12, fixed=None, missing_frac=0.05),
    # (enough blocks for every rank of an 8-way sharding to hold some)
    'tiny8': dict(P=2, n_ld=8_000, B=40, M=12, fixed=None, missing_frac=0.05),
    # C4: as C3 but eigen-truncated LD (what --ldthresh 0.8 leaves: kept rank ~0.28 n)
    'C4': dict(P=2, n_ld=1_000_000, B=1700, M=40, fixed=None, missing_frac=0.05,
               kind='lowrank', rank_frac=0.28),
    # C5: 4 cohorts, 5M SNPs in 8500 LDetect-sized blocks, M=81: ~70 GB of
    # LD panels + 2 x 13.6 GB of vi_mu resident on one GPU
    'C5': dict(P=4, n_ld=5_000_000, B=8500, M=81, fixed=None, missing_frac=0.05),
    'tiny4': dict(P=2, n_ld=6_000, B=12, M=12, fixed=None, missing_frac=0.05,
                  kind='lowrank', rank_frac=0.28),
    # C4f: C4 by SURVEY.md 8d's recipe to the letter -- per block and cohort the factor model
    # R = D^-1/2 (F F^T / m + 0.05 I) D^-1/2, F ~ N(0,1)^{n x m}, m = ceil(n / 4), eigendecomposed
    # and cut by the loader's --ldthresh 0.8 rule (kept rank = m or a few more: the F F^T part
    # sits above 1, the 0.05 floor below 1 - sqrt(0.8) except where a small block's diagonal
    # scaling lifts it).  Setup eigendecomposes 3 400 blocks on the GPU.
    'C4f': dict(P=2, n_ld=1_000_000, B=1700, M=40, fixed=None, missing_frac=0.05,
                kind='lowrank', spectrum='factor', rank_frac=0.25),
    'tiny4f': dict(P=2, n_ld=6_000, B=12, M=12, fixed=None, missing_frac=0.05,
                   kind='lowrank', spectrum='factor', rank_frac=0.25),
    # C3K12: C3's SNPs and LD under the mixture `vilma fit` builds BY DEFAULT for two cohorts: the
    # grid of _make_simple at -K 12 (reference vi_options.py:17-20, 284-337) = 582 components
    # (SURVEY.md section 0.4).  The per-SNP passes then move 14.5x C3's bytes and dominate the sweep.
    'C3K12': dict(P=2, n_ld=1_000_000, B=1700, M=582, fixed=None, missing_frac=0.05,
                  mixture='make_simple', K=12),
    'tinyK12': dict(P=2, n_ld=6_000, B=12, M=582, fixed=None, missing_frac=0.05,
                    mixture='make_simple', K=12),
}
FACTOR_LD_THRESH = 0.8


def block_sizes(n_ld, B, fixed=None, seed=0):
    """LDetect-like block sizes: lognormal(ln 520, 0.5) clipped to [50, 3000],
    rescaled to sum to n_ld, residual spread +-1 over blocks in descending-size order."""
    if fixed is not None:
        assert fixed * B == n_ld
        return np.full(B, fixed, dtype=np.int64)
    rng = np.random.default_rng(seed)
    n = np.clip(np.round(rng.lognormal(np.log(520), 0.5, B)), 50, 3000)
    n = np.clip(np.round(n * n_ld / n.sum()), 50, 3000).astype(np.int64)
    resid = int(n_ld - n.sum())
    order = np.argsort(-n, kind='stable')
    i = 0
    while resid != 0:
        b = order[i % B]
        step = 1 if resid > 0 else -1
        if 50 <= n[b] + step <= 3000:
            n[b] += step
            resid -= step
        i += 1
    return n


def missing_counts(sizes, missing_frac, seed=0):
    """LD-missing SNPs attached after each block; missing_frac is of ALL SNPs."""
    if missing_frac <= 0:
        return np.zeros(len(sizes), dtype=np.int64)
    n_ld = int(sizes.sum())
    total = int(round(n_ld * missing_frac / (1.0 - missing_frac)))
    rng = np.random.default_rng([seed, 17])
    return rng.multinomial(total, sizes / sizes.sum()).astype(np.int64)


def mixture_covs(P, M):
    """sigma_k^2 = geomspace(1e-8, 1e-2, M); cross-cohort correlation cycling 0, 0.5, 0.9."""
    var = np.geomspace(1e-8, 1e-2, M)
    covs = []
    for k in range(M):
        r = 0.0 if P == 1 else (0.0, 0.5, 0.9)[k % 3]
        covs.append(var[k] * ((1 - r) * np.eye(P) + r * np.ones((P, P))))
    return np.array(covs)


def make_simple_covs(P, K, seed):
    """The reference's default mixture grid (vi_options._make_simple, reference vi_options.py:
    284-337) for squared effects between 1e-7 and 2e-3 in every cohort -- the range its data-driven
    heuristic (:196-226) lands in for sumstats of this recipe.  The grid draws from numpy's global
    legacy RNG; the caller's stream is put back afterwards."""
    from .vi_options import _make_simple
    state = np.random.get_state()
    try:
        np.random.seed(100003 + seed)
        covs = _make_simple(P, K, np.full(P, 1e-7), np.full(P, 2e-3))
    finally:
        np.random.set_state(state)
    return np.array(covs, dtype=np.float64)


def shard_ranges(sizes, miss, world, per_snp_cost):
    """Contiguous runs of blocks per rank, balanced by 8 n_b^2 + per-SNP cost."""
    cost = 8.0 * sizes.astype(np.float64) ** 2 + per_snp_cost * (sizes + miss)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, cum[-1] * r / world)))
    cuts.append(len(sizes))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class BlockData:
    """The random content of one block (all cohorts), reproducible from (seed, block)."""

    def __init__(self, seed, b, n, m, P):
        rng = np.random.default_rng([seed, 1000 + b])
        self.n, self.m = int(n), int(m)
        self.rho = rng.uniform(0.5, 0.95, size=P)
        self.se = rng.uniform(0.005, 0.02, size=(P, self.n))
        causal = rng.random(self.n) < 0.05
        cov = 0.01 ** 2 * (0.2 * np.eye(P) + 0.8 * np.ones((P, P)))
        self.beta = np.zeros((P, self.n))
        if causal.any():
            self.beta[:, causal] = rng.multivariate_normal(np.zeros(P), cov,
                                                           size=int(causal.sum())).T
        self.eps = rng.normal(size=(P, self.n))            # sampling noise of beta-hat
        self.init_noise = rng.normal(size=(P, self.n))     # the 1e-3*se jitter of _initialize


def _progress(what, i, total, every=1000):
    """A line on stderr every `every` blocks: setup of the big workloads takes minutes and a silent
    job looks hung to a batch runner."""
    if i and i % every == 0:
        print('[synthetic] %s: block %d of %d' % (what, i, total), file=sys.stderr, flush=True)


def ar1_numpy(n, rho):
    idx = np.arange(n)
    return rho ** np.abs(idx[:, None] - idx[None, :])


class SyntheticShard:
    """One rank's part of a synthetic problem: per-SNP arrays in local SNP order, the local
    perm per cohort, and a generator of LD blocks (numpy or device tensors)."""

    def __init__(self, P, n_ld, B, M, fixed=None, missing_frac=0.0, seed=0, rank=0, world=1,
                 gwas_N=1e5, init_hg=0.1, block_range=None, kind='ar1', rank_frac=0.28,
                 spectrum='geom', mixture='geom', K=None):
        self.P, self.M, self.A, self.seed = P, M, 1, seed
        self.kind, self.rank_frac, self.spectrum = kind, rank_frac, spectrum
        self.gwas_N = np.full(P, gwas_N, dtype=np.float64)
        self.init_hg = np.full(P, init_hg, dtype=np.float64)
        self.sizes_all = block_sizes(n_ld, B, fixed, seed)
        self.miss_all = missing_counts(self.sizes_all, missing_frac, seed)
        self.N_global = int(self.sizes_all.sum() + self.miss_all.sum())
        self.n_ld_global = int(self.sizes_all.sum())
        if block_range is None:
            block_range = shard_ranges(self.sizes_all, self.miss_all, world,
                                       per_snp_cost=8.0 * 3 * M * P)[rank]
        self.b0, self.b1 = block_range
        if self.b1 <= self.b0:
            raise ValueError('rank %d of %d would hold no LD block of this workload (%d blocks)'
                             % (rank, world, len(self.sizes_all)))
        self.sizes = self.sizes_all[self.b0:self.b1]
        self.miss = self.miss_all[self.b0:self.b1]
        self.blocks = [BlockData(seed, self.b0 + i, n, m, P)
                       for i, (n, m) in enumerate(zip(self.sizes, self.miss))]
        self.N = int(self.sizes.sum() + self.miss.sum())
        self.n_ld = int(self.sizes.sum())
        # local SNP index of every LD position (block SNPs first, then that block's missing)
        snp_start = np.concatenate([[0], np.cumsum(self.sizes + self.miss)])
        ld_pos, miss_pos = [], []
        for i, (n, m) in enumerate(zip(self.sizes, self.miss)):
            ld_pos.append(np.arange(snp_start[i], snp_start[i] + n))
            miss_pos.append(np.arange(snp_start[i] + n, snp_start[i] + n + m))
        self.snp_start = snp_start
        self.ld_snps = np.concatenate(ld_pos).astype(np.int64)
        self.missing = (np.concatenate(miss_pos).astype(np.int64) if miss_pos
                        else np.zeros(0, dtype=np.int64))
        self.perm = np.concatenate([self.ld_snps, self.missing])
        if mixture == 'make_simple':
            self.covs = make_simple_covs(P, K, seed)
            assert len(self.covs) == M, (len(self.covs), M)
        else:
            self.covs = mixture_covs(P, M)
        if kind == 'ar1':
            self.ranks_all = self.sizes_all.astype(np.int64)
        elif spectrum == 'factor':
            # m = ceil(n / 4): the nominal kept rank; build() replaces the local entries by what
            # the threshold actually keeps, per cohort (ranks_by_cohort)
            self.ranks_all = -(-self.sizes_all.astype(np.int64) // 4)
        else:
            self.ranks_all = np.maximum(1, np.round(rank_frac * self.sizes_all)).astype(np.int64)
        self.ranks = self.ranks_all[self.b0:self.b1]
        self.ranks_by_cohort = [self.ranks.copy() for _ in range(P)]
        # algorithmic bytes of one product: 8 n^2 dense, 8 n r eigen form
        self.ld_bytes = 8 * int((self.sizes.astype(np.int64) * self.ranks).sum()) * P
        self._eig = None          # per (cohort, block): (U, s) device tensors, lowrank kind

    # ------------------------------------------------------------------ LD blocks
    def ld_blocks_numpy(self, p):
        for blk in self.blocks:
            yield ('dense', ar1_numpy(blk.n, blk.rho[p]))

    def ld_blocks_torch(self, p, device, form='auto'):
        import torch
        if self.kind == 'lowrank':
            from .matrix_structures import dense_is_cheaper
            for (U, sv) in self._eig[p]:
                n, r = U.shape
                if form == 'dense' or (form == 'auto' and dense_is_cheaper(n, r)):
                    yield ('dense', (U * sv) @ U.T)
                else:
                    yield ('eig', U, sv)
            return
        for blk in self.blocks:
            idx = torch.arange(blk.n, device=device, dtype=torch.float64)
            yield ('dense', torch.pow(float(blk.rho[p]), (idx[:, None] - idx[None, :]).abs()))

    def block_specs(self, form='auto', p=0):
        """(form, n, r) per local block of cohort p, in the order ld_blocks_torch(p, ...) yields."""
        if self.kind == 'lowrank':
            from .matrix_structures import dense_is_cheaper
            out = []
            for n, r in zip(self.sizes, self.ranks_by_cohort[p]):
                dense = form == 'dense' or (form == 'auto' and dense_is_cheaper(int(n), int(r)))
                out.append(('dense', int(n), int(n)) if dense else ('eig', int(n), int(r)))
            return out
        return [('dense', int(n), int(n)) for n in self.sizes]

    # ------------------------------------------------------------------ sumstats + constants
    def build(self, device=None):
        """Sumstats and the load-time constants for this shard (closed forms on the host for
        AR(1) LD; the eigen-form workload builds its factors on `device` with torch.linalg).
        Sets: betahat, se, ld_diags, sld, adj, scalings, annot [.., N] local arrays;
        chi_local [P], rank_local [P], inv_se2_local [P]; call finish_init(inv_se2_global) to
        get inverse_betas."""
        if self.kind == 'lowrank':
            return self._build_lowrank(device)
        P, N = self.P, self.N
        self.betahat = np.zeros((P, N))
        self.se = np.ones((P, N))
        self.ld_diags = np.zeros((P, N))
        self.ld_diags[:, self.ld_snps] = 1.0
        self.chi_local = np.zeros(P)
        self._z = np.zeros((P, N))
        for i, blk in enumerate(self.blocks):
            _progress('sumstats', i, len(self.blocks))
            lo = self.snp_start[i]
            sl = slice(lo, lo + blk.n)
            self.se[:, sl] = blk.se
            for p in range(P):
                zt = blk.beta[p] / blk.se[p]
                zhat, chi = _block_sumstats(blk.n, blk.rho[p], zt, blk.eps[p], device)
                self.betahat[p, sl] = blk.se[p] * zhat
                self._z[p, sl] = zhat
                self.chi_local[p] += chi
        self.rank_local = np.full(P, float(self.n_ld))
        self.sld = self.ld_diags / self.se ** 2
        self.adj = self._z / self.se            # (R R^-1 z)/se
        self.scalings = np.ones((P, N))
        self.annot = np.zeros(N, dtype=np.int32)
        self.inv_se2_local = (self.se ** -2).sum(axis=1)
        self._device = device
        return self

    # ------------------------------------------------------------------ eigen-form LD (C4)
    def _build_lowrank(self, device):
        """LD given directly in eigen form R = U diag(s) U^T with orthonormal U [n, r] (QR of a
        seeded Gaussian matrix) and a decaying spectrum with trace n -- the shape --ldthresh 0.8
        leaves.  Every load-time constant has a closed form in (U, s): R^+ = U s^-1 U^T,
        adj = U U^T z / se, chi = |U^T z|^2_{1/s}, rank = r, diag = sum_c s_c U_ic^2."""
        import torch
        assert device is not None, 'the eigen-form synthetic workload is built on the GPU'
        P, N = self.P, self.N
        self.betahat = np.zeros((P, N))
        self.se = np.ones((P, N))
        self.ld_diags = np.zeros((P, N))
        self.chi_local = np.zeros(P)
        self.adj = np.zeros((P, N))
        self._eig = [[] for _ in range(P)]
        f64 = dict(dtype=torch.float64, device=device)
        factor_eig = None
        if self.spectrum == 'factor':
            factor_eig = [_factor_model_eig_stacked(
                [(i, np.random.default_rng([self.seed, 5000 + self.b0 + i, p]), blk.n)
                 for i, blk in enumerate(self.blocks)], device) for p in range(P)]
        for i, blk in enumerate(self.blocks):
            lo = self.snp_start[i]
            sl = slice(lo, lo + blk.n)
            r = int(self.ranks[i])
            self.se[:, sl] = blk.se
            for p in range(P):
                rng = np.random.default_rng([self.seed, 5000 + self.b0 + i, p])
                if self.spectrum == 'factor':
                    U, sv = factor_eig[p].pop(i)
                    self.ranks_by_cohort[p][i] = U.shape[1]
                    r = U.shape[1]
                    rng.normal(size=(blk.n, -(-blk.n // 4)))      # F was drawn from this stream
                else:
                    G = torch.as_tensor(rng.normal(size=(blk.n, r)), **f64)
                    U = torch.linalg.qr(G)[0].contiguous()
                    sv = torch.as_tensor(np.geomspace(1.0, 0.05, r), **f64)
                    sv = sv * (blk.n / sv.sum())
                self._eig[p].append((U, sv))
                se = torch.as_tensor(blk.se[p], **f64)
                zt = torch.as_tensor(blk.beta[p] / blk.se[p], **f64)
                eps = torch.as_tensor(rng.normal(size=r), **f64)
                zhat = U @ (sv * (U.T @ zt)) + U @ (sv.sqrt() * eps)
                proj = U.T @ zhat
                self.chi_local[p] += float((proj * proj / sv).sum().item())
                self.betahat[p, sl] = (se * zhat).cpu().numpy()
                self.adj[p, sl] = ((U @ proj) / se).cpu().numpy()
                self.ld_diags[p, sl] = ((U * U) @ sv).cpu().numpy()
        self.rank_local = np.array([float(rk.sum()) for rk in self.ranks_by_cohort])
        self.ld_bytes = 8 * int(sum((self.sizes.astype(np.int64) * rk).sum()
                                    for rk in self.ranks_by_cohort))
        self.sld = self.ld_diags / self.se ** 2
        self.scalings = np.ones((P, N))
        self.annot = np.zeros(N, dtype=np.int32)
        self.inv_se2_local = (self.se ** -2).sum(axis=1)
        self._device = device
        return self

    def _finish_lowrank(self, prior):
        import torch
        P, N = self.P, self.N
        f64 = dict(dtype=torch.float64, device=self._device)
        for i, blk in enumerate(self.blocks):
            lo = self.snp_start[i]
            sl = slice(lo, lo + blk.n)
            for p in range(P):
                U, sv = self._eig[p][i]
                se = torch.as_tensor(blk.se[p], **f64)
                dinv = 1.0 / (se * se / prior[p])                         # Woodbury on U s U^T + D
                rhs = torch.as_tensor(self.adj[p, sl] * blk.se[p], **f64)
                y = dinv * rhs
                core = torch.diag(1.0 / sv) + U.T @ (dinv[:, None] * U)
                sol = y - dinv * (U @ torch.linalg.solve(core, U.T @ y))
                self.inverse_betas[p, sl] = (sol * se).cpu().numpy()
                self.fake_mu[p, sl] = self.inverse_betas[p, sl] + 1e-3 * blk.se[p] * blk.init_noise[p]
        return self

    def finish_init(self, inv_se2_global):
        """Ridge start inverse_betas (variational_inference.py:246-252) and the jittered
        fake_mu of _initialize (:649-657), per block."""
        P, N = self.P, self.N
        prior = 2 * self.gwas_N * self.init_hg / np.asarray(inv_se2_global)
        self.inverse_betas = np.zeros((P, N))
        self.fake_mu = np.zeros((P, N))
        if self.kind == 'lowrank':
            return self._finish_lowrank(prior)
        for i, blk in enumerate(self.blocks):
            _progress('ridge start', i, len(self.blocks))
            lo = self.snp_start[i]
            sl = slice(lo, lo + blk.n)
            for p in range(P):
                rhs = self.adj[p, sl] * blk.se[p]
                sol = _block_ridge(blk.n, blk.rho[p], blk.se[p] ** 2 / prior[p], rhs,
                                   self._device)
                self.inverse_betas[p, sl] = sol * blk.se[p]
                self.fake_mu[p, sl] = self.inverse_betas[p, sl] + 1e-3 * blk.se[p] * blk.init_noise[p]
        # LD-missing SNPs: NaN in every cohort -> cross-cohort nanmean is NaN -> 0 (:653-657)
        return self


def _factor_model_eig_stacked(blocks, device):
    """_factor_model_eig for many blocks at once: (key, rng, n) triples -> {key: (U, s)}.  The
    matrices are eigendecomposed in stacks like the loader's (ld_device.gpu_stacks: rocSOLVER's
    batched solver is launch-bound, a stack costs about what one matrix does)."""
    import torch
    from . import ld_device
    from .matrix_structures import select_eigenpairs
    sizes = {key: n for key, _, n in blocks}
    rngs = {key: rng for key, rng, _ in blocks}
    out = {}
    stacks = ld_device.gpu_stacks(sizes, list(sizes))
    for si, members in enumerate(stacks):
        _progress('factor-model eigh, stack', si, len(stacks), every=20)
        npad = ld_device._padded(max(sizes[k] for k in members))
        stack = torch.zeros((len(members), npad, npad), dtype=torch.float64, device=device)
        for j, key in enumerate(members):
            n = sizes[key]
            m = -(-n // 4)
            F = torch.as_tensor(rngs[key].normal(size=(n, m)), dtype=torch.float64, device=device)
            R = F @ F.T / m
            R.diagonal().add_(0.05)
            d = R.diagonal().rsqrt()
            stack[j, :n, :n] = d[:, None] * R * d[None, :]
            if n < npad:
                stack[j].diagonal()[n:] = -(2.0 * n + 1.0)        # below -sum_j |R_ij| >= -n
        w, V = torch.linalg.eigh(stack)
        w_host = w.cpu().numpy()
        for j, key in enumerate(members):
            n = sizes[key]
            pad, m = npad - n, -(-n // 4)
            keep, degenerate = select_eigenpairs(w_host[j, pad:], FACTOR_LD_THRESH)
            if degenerate is not None or keep.size < m:
                raise RuntimeError('factor-model block of %d SNPs kept %d eigenpairs, expected '
                                   '>= %d' % (n, keep.size, m))
            idx = torch.as_tensor(keep + pad, device=device)
            out[key] = (V[j, :n, idx].contiguous(), w[j, idx].contiguous())
    return out


def _factor_model_eig(rng, n, m, device):
    """R = D^-1/2 (F F^T / m + 0.05 I) D^-1/2 with F ~ N(0,1)^{n x m}, then what
    the loader keeps at --ldthresh 0.8 (select_eigenpairs = the reference's _svd_threshold +
    LowRankMatrix rules).  Returns (U [n, r], s [r]) on `device`; r >= m (the F F^T part is
    always kept; a few floor eigenvalues of a small block can pass the threshold too)."""
    import torch
    from .matrix_structures import select_eigenpairs
    F = torch.as_tensor(rng.normal(size=(n, m)), dtype=torch.float64, device=device)
    R = F @ F.T / m
    R.diagonal().add_(0.05)
    d = R.diagonal().rsqrt()
    R = d[:, None] * R * d[None, :]
    w, V = torch.linalg.eigh(R)
    keep, degenerate = select_eigenpairs(w.cpu().numpy(), FACTOR_LD_THRESH)
    if degenerate is not None or keep.size < m:
        raise RuntimeError('factor-model block of %d SNPs kept %d eigenpairs, expected >= %d'
                           % (n, keep.size, m))
    idx = torch.as_tensor(keep, device=device)
    return V[:, idx].contiguous(), w[idx].contiguous()


def ar1_apply(rho, x):
    """R x for the AR(1) matrix R_ij = rho^|i-j| by its O(n) two-sided recursion."""
    from scipy.signal import lfilter
    fwd = lfilter([1.0], [1.0, -rho], x)
    bwd = lfilter([1.0], [1.0, -rho], x[::-1])[::-1]
    return fwd + bwd - x


def _block_sumstats(n, rho, z_true, eps, device=None):
    """z-hat = R z + R^(1/2) eps (recipe of reference sim.py:136-156) and z-hat^T R^-1 z-hat,
    in closed form for an AR(1) block: the Cholesky factor of R is the AR(1) recursion itself
    (L eps: w_0 = eps_0, w_i = rho w_{i-1} + sqrt(1-rho^2) eps_i), so everything is O(n) on the
    host -- the 34 000 block-cohorts of workload C5 need no per-block factorisation."""
    from scipy.signal import lfilter
    c = np.sqrt(1.0 - rho * rho)
    e = c * eps
    e[0] = eps[0]
    zhat = ar1_apply(rho, z_true) + lfilter([1.0], [1.0, -rho], e)
    w = np.empty(n)                       # w = L^-1 z-hat
    w[0] = zhat[0]
    w[1:] = (zhat[1:] - rho * zhat[:-1]) / c
    return zhat, float(w @ w)


def _block_ridge(n, rho, reg, rhs, device=None):
    """(R + diag(reg))^-1 rhs for an AR(1) block: R^-1 = T is tridiagonal, so
    (I + T diag(reg)) x = T rhs is one banded solve."""
    if n == 1:
        return rhs / (1.0 + reg)
    from scipy.linalg import solve_banded
    q = 1.0 / (1.0 - rho * rho)
    tdiag = np.full(n, (1.0 + rho * rho) * q)
    tdiag[0] = tdiag[-1] = q
    toff = -rho * q
    b = tdiag * rhs
    b[:-1] += toff * rhs[1:]
    b[1:] += toff * rhs[:-1]
    ab = np.zeros((3, n))
    ab[0, 1:] = toff * reg[1:]
    ab[1] = 1.0 + tdiag * reg
    ab[2, :-1] = toff * reg[:-1]
    return solve_banded((1, 1), ab, b)
