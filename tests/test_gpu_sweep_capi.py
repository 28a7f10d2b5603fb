"""The sweep-level C-ABI (vilma_sweep, vilma_elbo, vilma_posterior, vilma_set_state /
vilma_get_state) driven through ctypes ALONE -- no MultiPopVI, no SweepDriver, no torch tensors, the
NULL stream -- reproduces every trajectory the reference recorded (tests/golden/traj_*.npz, written
by tests/golden/make_golden.py from /root/reference/src/vilma/variational_inference.py:353-389):
L bit-equal, ELBO 1e-9 relative, posterior means 1e-7.  This is the fit a non-Python host would run."""
import ctypes as C

import numpy as np
import pytest

from helpers import golden, traj_blocks, TRAJ_NAMES

pytestmark = pytest.mark.gpu


_KEEP = []       # arrays whose addresses were handed to the library stay alive for the test run


def _p(a):
    _KEEP.append(a)
    return C.c_void_p(a.ctypes.data)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CFit:
    """A fit set up and run with nothing but the entry points of include/vilma_hip.h."""

    def __init__(self, g, form='dense'):
        from vilma_amd import _lib
        from oracle.ldop import EigenBlock
        self.lib = lib = _lib.load()
        self.g = g
        P, N, M = int(g['P']), int(g['N']), len(g['covs'])
        A = g['annotations'].shape[1]
        self.P, self.N, self.M, self.A = P, N, M, A
        ctx = C.c_void_p()
        assert lib.vilma_create(P, N, M, A, C.byref(ctx)) == 0, lib.vilma_last_error(None)
        self.ctx = ctx
        se = _f64(g['se'])
        if bool(g['scaled']):
            se = np.ones_like(se)
        self.se = se
        adj, sld = _f64(g['adj_marginal_effects']), _f64(se ** -2 * g['ld_diags'])
        annot = np.ascontiguousarray(np.where(g['annotations'])[1], dtype=np.int32)
        self.ok(lib.vilma_set_snp_data(ctx, _p(adj), _p(se), _p(sld), _p(_f64(g['scalings'])),
                                       _p(annot)))
        prec = _f64(g['mixture_prec']).reshape(M, P, P)
        self.ok(lib.vilma_set_mixture(ctx, _p(prec), _p(_f64(g['log_det']))))
        self.ok(lib.vilma_set_annotation_counts(ctx, _p(_f64(g['annotations'].sum(axis=0)))))
        perm = np.ascontiguousarray(g['perm'], dtype=np.int64)
        n_ld = int(N - len(g['missing']))
        t = float(g['ldthresh'])
        for p, blocks in enumerate(traj_blocks(g)):
            eig = [EigenBlock(X, t) for X in blocks]       # the reference's thresholded factors
            recs = [('dense', _f64((b.u * b.s) @ b.u.T)) if form == 'dense'
                    else ('eig', _f64(b.u), _f64(b.s)) for b in eig]
            total = sum(lib.vilma_ld_dense_elems(r[1].shape[0]) if r[0] == 'dense'
                        else lib.vilma_ld_lowrank_elems(*r[1].shape) for r in recs)
            self.ok(lib.vilma_ld_begin(ctx, p, len(recs), n_ld, _p(perm), total))
            for r in recs:
                if r[0] == 'dense':
                    self.ok(lib.vilma_ld_add_dense(ctx, p, r[1].shape[0], _p(r[1])))
                else:
                    self.ok(lib.vilma_ld_add_lowrank(ctx, p, r[1].shape[0], r[1].shape[1],
                                                     _p(r[1]), _p(r[2])))
            self.ok(lib.vilma_ld_end(ctx, p))
        self.ok(lib.vilma_set_fit_constants(ctx, _p(_f64(g['chi_stat'])), _p(_f64(g['ld_ranks'])),
                                            1 if bool(g['scale_se']) else 0))

    def ok(self, rc):
        assert rc == 0, self.lib.vilma_last_error(self.ctx).decode()

    def fake_mu(self):
        """The host's part of _initialize (variational_inference.py:643-657): one legacy-RNG draw."""
        g, P, N = self.g, self.P, self.N
        missing = np.isclose(g['ld_diags'], 0)
        np.random.seed(int(g['seed']))
        fake = np.random.normal(loc=np.copy(g['inverse_betas']), scale=1e-3 * self.se, size=(P, N))
        fake[missing] = np.nan
        n_obs = (~missing).sum(axis=0)
        col = np.where(n_obs > 0, np.nansum(fake, axis=0) / np.maximum(n_obs, 1), np.nan)
        fill = np.tile(col, [P, 1])
        fake[missing] = fill[missing]
        fake[np.isnan(fake)] = 0.
        return _f64(fake)

    def close(self):
        self.lib.vilma_destroy(self.ctx)


def _reproduce_trajectory(name, extra_flags=0):
    from vilma_amd import _lib
    g = golden('traj_%s.npz' % name)
    fit = CFit(g, form='eig' if name.endswith('lowrank') or name.endswith('_lr') else 'dense')
    lib, ctx = fit.lib, fit.ctx
    obj = C.c_double()
    fit.ok(lib.vilma_initialize(ctx, None, _p(fit.fake_mu()), C.byref(obj)))
    assert abs(obj.value - float(g['init_elbo'])) < 1e-9 * abs(obj.value)
    hyper = np.empty((fit.A, fit.M))
    fit.ok(lib.vilma_get_state(ctx, None, None, _p(hyper), None))
    np.testing.assert_allclose(hyper, g['init_hyper_delta'], rtol=1e-10)
    fit.ok(lib.vilma_snapshot_mean(ctx, None))
    L = np.ones(5)
    elbo, running = C.c_double(obj.value), C.c_double(float('nan'))
    stats = _lib.SweepStats()
    mean = np.empty((fit.P, fit.N))
    n = len(g['elbo'])
    for it in range(n):
        fit.ok(lib.vilma_sweep(ctx, None, _p(L), C.byref(elbo), C.byref(running), 2.0,
                               _lib.SWEEP_DIFF | (extra_flags if it + 1 < n else 0), C.byref(stats)))
        assert abs(elbo.value - g['elbo'][it]) < 1e-9 * abs(elbo.value), (it, elbo.value)
        assert np.array_equal(L, g['L'][it]), (it, L, g['L'][it])
        assert stats.n_evaluations <= int(g['objs_per_sweep'][it])
        np.testing.assert_allclose(stats.error_scaling[:fit.P], g['error_scaling'][it], rtol=1e-8)
        fit.ok(lib.vilma_elbo(ctx, C.byref(obj)))
        assert abs(obj.value - elbo.value) < 1e-9 * abs(obj.value)
        if it in (0, n - 1):
            fit.ok(lib.vilma_posterior(ctx, _p(mean), None))
            np.testing.assert_allclose(mean, g['post_mean'][it], rtol=1e-7, atol=1e-12)
    mu, delta = np.empty((fit.M, fit.P, fit.N)), np.empty((fit.N, fit.M))
    tau = np.empty(fit.P)
    fit.ok(lib.vilma_get_state(ctx, _p(mu), _p(delta), _p(hyper), _p(tau)))
    np.testing.assert_allclose(mu, g['final_vi_mu'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(delta, g['final_vi_delta'], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(hyper, g['hyper_delta'][n - 1], rtol=1e-6, atol=1e-300)
    var = np.empty((fit.P, fit.N))
    fit.ok(lib.vilma_posterior(ctx, None, _p(var)))
    np.testing.assert_allclose(var, g['final_post_var'], rtol=1e-7)
    # vilma_set_state with what vilma_get_state returned is the same state: same ELBO
    fit.ok(lib.vilma_set_state(ctx, None, _p(mu), _p(hyper), _p(tau), C.byref(obj)))
    assert abs(obj.value - elbo.value) < 1e-9 * abs(obj.value)
    fit.close()



@pytest.mark.parametrize('name', TRAJ_NAMES)
def test_vilma_sweep_reproduces_the_reference_trajectories(name):
    _reproduce_trajectory(name)


@pytest.mark.parametrize('flags', [0, 2], ids=['host-decided', 'queued-ahead'])
@pytest.mark.parametrize('name', ['p1_dense', 'p2_scale_se', 'p4_m81', 'p2_bigblock_lr'])
def test_debug_poison_changes_no_trajectory(name, flags, monkeypatch):
    """VILMA_DEBUG_POISON=1 (read by vilma_create): the result slots of both candidates and the
    candidates' vi_mu buffers are filled with NaN before every beta trial.  Whatever a trial is
    defined to produce it must then have written, and nothing may be read that was not produced:
    the reference's trajectories come out unchanged, L to the bit."""
    monkeypatch.setenv('VILMA_DEBUG_POISON', '1')
    _reproduce_trajectory(name, extra_flags=flags)


def test_debug_poison_reaches_an_unwritten_slot(monkeypatch):
    """The other half: with the switch on, candidate B's result slot after a ONE-step trial is NaN
    (nothing produced it), so a decision that read it would fail loudly."""
    from vilma_amd import _lib
    monkeypatch.setenv('VILMA_DEBUG_POISON', '1')
    monkeypatch.setenv('VILMA_TWO_STEP', '0')
    g = golden('traj_p1_dense.npz')
    fit = CFit(g)
    lib, ctx = fit.lib, fit.ctx
    obj = C.c_double()
    fit.ok(lib.vilma_initialize(ctx, None, _p(fit.fake_mu()), C.byref(obj)))
    L = np.ones(5)
    elbo, running = C.c_double(obj.value), C.c_double(float('nan'))
    stats = _lib.SweepStats()
    fit.ok(lib.vilma_sweep(ctx, None, _p(L), C.byref(elbo), C.byref(running), 2.0, 0, C.byref(stats)))
    assert abs(elbo.value - g['elbo'][0]) < 1e-9 * abs(elbo.value)      # the sweep itself is sound
    nt = 3 * fit.P + 2
    slot_b = np.zeros(nt)
    fit.ok(lib.vilma_debug_result_slot(ctx, 1, _p(slot_b), nt))
    assert np.all(np.isnan(slot_b)), slot_b
    fit.close()


@pytest.mark.parametrize('flags', [0, 2], ids=['host-decided', 'queued-ahead'])
def test_line_search_failure_is_the_reference_error(flags):
    """A NaN objective makes every comparison false: the search backs off until L > L_MAX and
    reports the reference's message (variational_inference.py:790-799) -- also when the sweep was
    promised ahead (VILMA_SWEEP_LOOKAHEAD): the device rejects step after step, gives up beyond
    L_MAX, and the host's line search, taking over where the device stood, raises the error."""
    from vilma_amd import _lib
    g = golden('traj_p1_dense.npz')
    fit = CFit(g)
    lib, ctx = fit.lib, fit.ctx
    # poison the problem: an infinite adjusted effect makes every objective non-finite
    adj = _f64(g['adj_marginal_effects']).copy()
    adj[0, 0] = np.inf
    annot = np.ascontiguousarray(np.where(g['annotations'])[1], dtype=np.int32)
    fit.ok(lib.vilma_set_snp_data(ctx, _p(adj), _p(fit.se), _p(_f64(fit.se ** -2 * g['ld_diags'])),
                                  _p(_f64(g['scalings'])), _p(annot)))
    obj = C.c_double()
    fit.ok(lib.vilma_initialize(ctx, None, _p(fit.fake_mu()), C.byref(obj)))
    assert not np.isfinite(obj.value)
    L = np.ones(5)
    elbo, running = C.c_double(obj.value), C.c_double(float('nan'))
    rc = lib.vilma_sweep(ctx, None, _p(L), C.byref(elbo), C.byref(running), 2.0, flags, None)
    assert rc != 0
    assert b'Encountered a numerical error.' in lib.vilma_last_error(ctx)
    fit.close()
