"""HipEngine: the device side of one shard of the VI problem, driven through the C-ABI.

The engine owns one `vilma_ctx` (include/vilma_hip.h).  torch is used only for the small
result vectors (so they can be all-reduced with torch.distributed / RCCL) and for the current
HIP stream; the LD store and the variational state live in memory the library allocates.
"""
import ctypes as C

import numpy as np

from . import _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class ResultLayout:
    """Slices of an engine's result vector (shared by the HIP engine and the test engine)."""

    def __init__(self, nt, am):
        self.nt, self.am = nt, am
        self.dsum = slice(0, 3)                        # convergence statistics, additive part
        self.totals = slice(3, 3 + nt)                 # sums of a plain evaluation (vilma_eval)
        self.ttotals = slice(3 + nt, 3 + 2 * nt)       # sums of a beta trial (candidate A)
        self.ttotals_b = slice(3 + 2 * nt, 3 + 3 * nt)  # ... of the second step of a two-step trial
        self.sums = slice(3 + 3 * nt, 3 + 3 * nt + am)  # responsibility sums (of candidate A)
        self.sums_b = slice(3 + 3 * nt + am, 3 + 3 * nt + 2 * am)   # ... of candidate B
        self.dmax = slice(3 + 3 * nt + 2 * am, 6 + 3 * nt + 2 * am)
        self.hyper = slice(6 + 3 * nt + 2 * am, 6 + 3 * nt + 3 * am)
        self.size = 6 + 3 * nt + 3 * am


class _DeviceVector:
    """A device buffer owned by the library, presented to torch without a copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8',
                                         'data': (int(ptr), False), 'version': 2}


class HipEngine:
    """One GPU, one shard: P cohorts x N SNPs x M mixture components, A annotations."""

    def __init__(self, P, N, M, A, device=None):
        import torch
        if not torch.cuda.is_available():
            raise _lib.VilmaHipError('no GPU visible to this process: the vilma fit hot path '
                                     'runs only on the HIP device (no CPU fallback)')
        self.torch = torch
        self.lib = _lib.load()
        if device is not None:
            torch.cuda.set_device(device)
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.P, self.N, self.M, self.A = int(P), int(N), int(M), int(A)
        ctx = C.c_void_p()
        if self.lib.vilma_create(self.P, self.N, self.M, self.A, C.byref(ctx)):
            raise _lib.VilmaHipError(self.lib.vilma_last_error(None).decode())
        self.ctx = ctx
        self.refresh_stream()
        # One device vector for every small result (owned by the context, include/vilma_hip.h), so
        # a decision needs a single D2H copy and the parts summed over ranks are contiguous:
        #   [diff sums (3) | eval totals (3P+2) | trial totals A (3P+2) | trial totals B (3P+2) |
        #    delta sums A (A*M) | delta sums B (A*M) | diff maxima (3) | hyper (A*M)]
        nt, am = _lib.ntotals(self.P), self.A * self.M
        self.layout = ResultLayout(nt, am)
        if self.lib.vilma_results_size(self.ctx) != self.layout.size:
            raise _lib.VilmaHipError('result vector layouts of the library and the binding differ')
        self._results_ptr = self.lib.vilma_results_dev(self.ctx)
        self.results = torch.as_tensor(_DeviceVector(self._results_ptr, self.layout.size),
                                       device=self.device)
        L = self.layout
        self._dsum = self.results[L.dsum]
        self._totals = self.results[L.totals]
        self._ttotals = self.results[L.ttotals]
        self._ttotals_b = self.results[L.ttotals_b]
        self._sums = self.results[L.sums]
        self._sums_b = self.results[L.sums_b]
        self._dmax = self.results[L.dmax]
        self._hyper = self.results[L.hyper]
        # device addresses of the fixed result slices, resolved once (a data_ptr() lookup per
        # launch costs about a microsecond of host time on the decision path)
        self._p = {name: C.c_void_p(t.data_ptr()) for name, t in (
            ('results', self.results), ('dsum', self._dsum), ('totals', self._totals),
            ('ttotals', self._ttotals), ('ttotals_b', self._ttotals_b), ('sums', self._sums), ('sums_b', self._sums_b), ('dmax', self._dmax),
            ('hyper', self._hyper))}
        self.n_totals = nt
        self._host = np.zeros(L.size)
        self._host_ptr = _ptr(self._host)
        self._sweep_box = None
        self._comm_cb = None        # keeps the ctypes callback of bind_comm alive
        self.collective = 'none'

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc:
            raise _lib.VilmaHipError(self.lib.vilma_last_error(self.ctx).decode())

    def _stream(self):
        return self._stream_handle

    def refresh_stream(self):
        """Bind the engine to torch's current HIP stream (kernels, RCCL all-reduces issued through
        torch and the fetch then share one ordering).  Called at construction and by the driver
        at the start of every sweep; looking the stream up per launch costs microseconds."""
        self._stream_handle = C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.vilma_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self.torch.cuda.synchronize()

    # ------------------------------------------------------------------ static data
    def set_snp_data(self, adj, se, sld, scalings, annot):
        adj, se, sld, scalings = _f64(adj), _f64(se), _f64(sld), _f64(scalings)
        annot = np.ascontiguousarray(annot, dtype=np.int32)
        assert adj.shape == (self.P, self.N) and annot.shape == (self.N,)
        self._check(self.lib.vilma_set_snp_data(self.ctx, _ptr(adj), _ptr(se), _ptr(sld),
                                                _ptr(scalings), _ptr(annot)))

    def set_mixture(self, prec, log_det):
        prec, log_det = _f64(prec).reshape(self.M, self.P, self.P), _f64(log_det)
        self._check(self.lib.vilma_set_mixture(self.ctx, _ptr(prec), _ptr(log_det)))

    def set_tau(self, tau):
        tau = _f64(tau)
        self._check(self.lib.vilma_set_tau(self.ctx, _ptr(tau)))

    def set_annotation_counts(self, counts):
        counts = _f64(counts)
        assert counts.shape == (self.A,)
        self._check(self.lib.vilma_set_annotation_counts(self.ctx, _ptr(counts)))

    def mstep(self, sums=None):
        """Device M-step from (all-reduced) delta sums; returns the device view of hyper_delta."""
        src = self._p['sums'] if sums is None else C.c_void_p(sums.data_ptr())
        self._check(self.lib.vilma_mstep(self.ctx, self._stream_handle, src, self._p['hyper']))
        return self._hyper

    def set_hyper(self, hyper):
        hyper = _f64(hyper).reshape(self.A, self.M)
        self._check(self.lib.vilma_set_hyper(self.ctx, _ptr(hyper)))

    # ------------------------------------------------------------------ LD store
    def load_ld(self, cohort, blocks, perm, n_ld, specs=None):
        """blocks: ('dense', R[n,n]) or ('eig', U[n,r], s[r]) per block in LD order, as a list or
        -- with `specs` = [(form, n, r), ...] given up front -- any iterable (so large synthetic
        LD can be produced block by block).  Arrays may be numpy (host) or torch CUDA tensors."""
        lib = self.lib
        if specs is None:
            blocks = list(blocks)
            specs = [(b[0], int(b[1].shape[0]), int(b[1].shape[-1])) for b in blocks]
        total = 0
        for form, n, r in specs:
            total += (lib.vilma_ld_dense_elems(n) if form == 'dense'
                      else lib.vilma_ld_lowrank_elems(n, r))
        self.ld_begin(cohort, len(specs), perm, n_ld, total)
        for b in blocks:
            self.ld_add(cohort, b)
        self.ld_end(cohort)

    def ld_begin(self, cohort, n_blocks, perm, n_ld, total_elems):
        """Start a cohort's LD store with room for `total_elems` doubles (an upper bound is fine:
        the streaming loader reserves before the ranks are known)."""
        perm = np.ascontiguousarray(perm, dtype=np.int64)
        assert perm.shape == (self.N,)
        self._check(self.lib.vilma_ld_begin(self.ctx, cohort, int(n_blocks), int(n_ld),
                                            _ptr(perm), int(total_elems)))

    def ld_add(self, cohort, b):
        if b[0] == 'dense':
            R = self._as_block(b[1])
            self._check(self.lib.vilma_ld_add_dense(self.ctx, cohort, int(b[1].shape[0]), R[1]))
        else:
            U = self._as_block(b[1])
            s = self._as_block(b[2])
            self._check(self.lib.vilma_ld_add_lowrank(self.ctx, cohort, int(b[1].shape[0]),
                                                      int(b[1].shape[1]), U[1], s[1]))

    def ld_end(self, cohort):
        self._check(self.lib.vilma_ld_end(self.ctx, cohort))

    def _as_block(self, a):
        if isinstance(a, np.ndarray):
            a = _f64(a)
            return a, _ptr(a)
        a = a.contiguous()      # torch tensor (device or host), float64
        assert a.dtype == self.torch.float64
        if a.is_cuda:
            # the library copies with hipMemcpy on the null stream: make sure whatever produced
            # the tensor on torch's current stream has finished
            self.torch.cuda.current_stream().synchronize()
        return a, C.c_void_p(a.data_ptr())

    def ld_bytes(self):
        alg, stored = C.c_int64(), C.c_int64()
        self._check(self.lib.vilma_ld_bytes(self.ctx, C.byref(alg), C.byref(stored)))
        return alg.value, stored.value

    def ld_matvec(self, x, cohort=-1):
        """BlockDiagonalMatrix.dot for all cohorts (x [P,N]) -> numpy [P,N]."""
        t = self.torch
        xd = t.as_tensor(_f64(x).reshape(self.P, self.N), device=self.device)
        yd = t.zeros_like(xd)
        self.ld_matvec_device(xd, yd, cohort)
        return yd.cpu().numpy()

    def ld_tile(self):
        """(rows per row strip, slabs per column strip, work items per launch) of the dense product."""
        v = [C.c_int(), C.c_int(), C.c_int()]
        self._check(self.lib.vilma_prof_ld_tile(self.ctx, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def ld_matvec2(self, xa, xb, cohort=-1):
        """Two right-hand sides in one pass over the LD store -> (numpy [P,N], numpy [P,N])."""
        t = self.torch
        xs = [t.as_tensor(_f64(x).reshape(self.P, self.N), device=self.device) for x in (xa, xb)]
        ys = [t.zeros_like(x) for x in xs]
        self._check(self.lib.vilma_ld_matvec2(self.ctx, self._stream(), cohort,
                                              C.c_void_p(xs[0].data_ptr()), C.c_void_p(xs[1].data_ptr()),
                                              C.c_void_p(ys[0].data_ptr()), C.c_void_p(ys[1].data_ptr())))
        return ys[0].cpu().numpy(), ys[1].cpu().numpy()

    def ld_matvec_device(self, xd, yd, cohort=-1):
        self._check(self.lib.vilma_ld_matvec(self.ctx, self._stream(), cohort,
                                             C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr())))

    # ------------------------------------------------------------------ state
    def set_mu(self, vi_mu):
        vi_mu = _f64(vi_mu)
        assert vi_mu.shape == (self.M, self.P, self.N)
        self._check(self.lib.vilma_set_mu(self.ctx, _ptr(vi_mu)))

    def get_mu(self):
        out = np.empty((self.M, self.P, self.N))
        self._check(self.lib.vilma_get_mu(self.ctx, _ptr(out)))
        return out

    def get_delta(self):
        out = np.empty((self.N, self.M))
        self._check(self.lib.vilma_get_delta(self.ctx, _ptr(out)))
        return out

    def get_vi_sigma(self, error_scaling=None):
        """[M, P, P, N] at `error_scaling` (None: the context's), formed on the device (outputs only)."""
        out = np.empty((self.M, self.P, self.P, self.N))
        tau = None if error_scaling is None else _f64(error_scaling)
        self._check(self.lib.vilma_get_vi_sigma(self.ctx, None if tau is None else _ptr(tau), _ptr(out)))
        return out

    def get_moments(self):
        mean, var = np.empty((self.P, self.N)), np.empty((self.P, self.N))
        self._check(self.lib.vilma_get_moments(self.ctx, _ptr(mean), _ptr(var)))
        return mean, var

    def init_state(self, fake_mu):
        """_initialize's per-SNP part on the device: vi_mu becomes the current state; returns the
        device view of the heuristic responsibility sums [A*M] (the `sums` slice of `results`)."""
        fake_mu = _f64(fake_mu)
        assert fake_mu.shape == (self.P, self.N)
        self._check(self.lib.vilma_init_state(self.ctx, self._stream_handle, _ptr(fake_mu),
                                              self._p['sums']))
        return self._sums

    def eval_given_delta(self, vi_delta):
        """Sums of (current vi_mu, the vi_delta given [N,M], current hyper/tau) plus
        max |vi_delta - derived delta| as a device tensor [3P+3]; the moments of that point are
        then available from get_trial_moments()."""
        t = self.torch
        vi_delta = np.asarray(vi_delta, dtype=np.float64)
        assert vi_delta.shape == (self.N, self.M)
        dkm = t.as_tensor(np.ascontiguousarray(vi_delta.T), device=self.device)
        out = t.zeros(self.n_totals + 1, dtype=t.float64, device=self.device)
        self._check(self.lib.vilma_eval_given_delta(self.ctx, self._stream_handle,
                                                    C.c_void_p(dkm.data_ptr()),
                                                    C.c_void_p(out.data_ptr())))
        self.torch.cuda.current_stream().synchronize()      # dkm must outlive the kernels
        return out

    def get_trial_moments(self):
        mean, var = np.empty((self.P, self.N)), np.empty((self.P, self.N))
        self._check(self.lib.vilma_get_trial_moments(self.ctx, _ptr(mean), _ptr(var)))
        return mean, var

    # ------------------------------------------------------------------ evaluations
    def eval(self, diff=False):
        """Evaluate the current vi_mu; with diff=True the convergence statistics against the
        snapshot ride in the same pass (results in the dsum / dmax slices) and the evaluated
        means become the snapshot -- only for evaluations accepted unconditionally."""
        if diff:
            self._check(self.lib.vilma_eval_diff(self.ctx, self._stream_handle, self._p['totals'],
                                                 self._p['dsum'], self._p['dmax']))
        else:
            self._check(self.lib.vilma_eval(self.ctx, self._stream_handle, self._p['totals']))
        return self._totals

    def trial(self, step):
        self._check(self.lib.vilma_trial_beta(self.ctx, self._stream_handle, float(step),
                                              self._p['ttotals']))
        return self._ttotals

    def trial2(self, step_a, step_b):
        """A beta trial at two step sizes in one pass over vi_mu and the LD store (candidates A
        and B; accept(1) / accept(2))."""
        self._check(self.lib.vilma_trial_beta2(self.ctx, self._stream_handle, float(step_a),
                                               float(step_b), self._p['ttotals'],
                                               self._p['ttotals_b']))
        return self._ttotals, self._ttotals_b

    def accept(self, take_mu):
        """0 / False: the last eval's moments; 1 / True: trial candidate A; 2: candidate B."""
        self._check(self.lib.vilma_accept(self.ctx, int(take_mu)))

    def fetch(self):
        """The whole result vector on the host (one pinned D2H copy behind the current stream)."""
        self._check(self.lib.vilma_fetch(self.ctx, self._stream_handle, self._p['results'],
                                         self._host_ptr, self.layout.size))
        return self._host.copy()

    def delta_sums(self, which=_lib.STATE_CURRENT):
        self._check(self.lib.vilma_delta_sums(self.ctx, self._stream_handle, self._p['sums'],
                                              which))
        return self._sums

    def trial_sums(self, both=False):
        """Responsibility sums of the last trial's candidate A (and B) from the per-tile sums its
        per-SNP pass left behind -- no second pass over vi_mu (vilma_trial_sums)."""
        self._check(self.lib.vilma_trial_sums(self.ctx, self._stream_handle, self._p['sums'],
                                              self._p['sums_b'] if both else None))
        return (self._sums, self._sums_b) if both else self._sums

    def trial_sums_available(self):
        return int(self.lib.vilma_trial_sums_available(self.ctx))

    def mean_diff(self):
        self._check(self.lib.vilma_mean_diff(self.ctx, self._stream_handle, self._p['dsum'],
                                             self._p['dmax']))
        return self._dsum, self._dmax       # slices of `results` (fetched with everything else)

    def snapshot_mean(self):
        self._check(self.lib.vilma_snapshot_mean(self.ctx, self._stream()))

    # ------------------------------------------------------------------ the sweep behind one call
    def set_fit_constants(self, chi_stat, ld_ranks, scale_se):
        chi, ranks = _f64(chi_stat), _f64(ld_ranks)
        assert chi.shape == (self.P,) and ranks.shape == (self.P,)
        self._check(self.lib.vilma_set_fit_constants(self.ctx, _ptr(chi), _ptr(ranks),
                                                     1 if scale_se else 0))

    def bind_comm(self, comm):
        """Give the context the collective its sweeps all-reduce with.  backend nccl: a
        communicator the context owns (ncclCommInitRank; the unique id travels through
        torch.distributed); anything else (gloo rehearsals): a callback into comm."""
        if comm is None or not comm.active:
            self._check(self.lib.vilma_comm_set_callback(
                self.ctx, _lib.ALLREDUCE_FN(), None, 1, 0))
            self.collective = 'none'
            return
        if comm.backend == 'nccl' and not comm.force_callback:
            # every rank must end up with the same kind of collective: agree on the outcome
            from .sharding import agree_on_rccl

            def make_id():
                ident = C.create_string_buffer(128)
                return ident.raw if self.lib.vilma_comm_unique_id(ident) == 0 else None

            def init_rccl(raw):
                self._check(self.lib.vilma_comm_init_rccl(self.ctx, comm.world, comm.rank,
                                                          C.create_string_buffer(raw, 128)))

            ok, err, failed = agree_on_rccl(comm, make_id, init_rccl)
            if ok:
                self.collective = 'rccl (communicator owned by the context)'
                return
            import logging
            logging.warning('the context could not set up its own RCCL communicator on %d rank(s) '
                            '(%s); the sweep\'s all-reduces go through torch.distributed instead',
                            failed, err)
        base, results = int(self._results_ptr), self.results

        def allreduce(_user, _stream, buf, n, op):
            try:
                off = (int(buf) - base) // 8
                if off < 0 or off + n > results.numel():
                    return 1
                comm.allreduce_inplace(results[off:off + n], op='max' if op else 'sum')
                return 0
            except Exception:       # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._comm_cb = _lib.ALLREDUCE_FN(allreduce)
        self._check(self.lib.vilma_comm_set_callback(self.ctx, self._comm_cb, None, comm.world,
                                                     comm.rank))
        self.collective = 'callback into torch.distributed (%s)' % comm.backend

    def comm_check(self):
        """All-reduce a one per rank through the context's collective on the engine's stream:
        the number of ranks that took part (1 without a collective)."""
        t = self.torch
        one = t.ones(1, dtype=t.float64, device=self.device)
        kind = C.c_int()
        self._check(self.lib.vilma_comm_info(self.ctx, C.byref(kind), None, None))
        if kind.value == 2:                 # the callback only knows slices of the result vector
            keep = self.results[:1].clone()
            self.results[:1] = 1.0
            self._check(self.lib.vilma_comm_allreduce(self.ctx, self._stream_handle,
                                                      self._p['results'], 1, 0))
            got = float(self.results[0].item())
            self.results[:1] = keep
            return int(round(got))
        self._check(self.lib.vilma_comm_allreduce(self.ctx, self._stream_handle,
                                                  C.c_void_p(one.data_ptr()), 1, 0))
        self.torch.cuda.current_stream().synchronize()
        return int(round(float(one.item())))

    def set_state(self, vi_mu, hyper, tau=None):
        """Make (vi_mu [M,P,N] or None = keep the device's, hyper_delta, error_scaling) the
        current state and evaluate it; returns its ELBO."""
        hyper = _f64(hyper).reshape(self.A, self.M)
        mu = None
        if vi_mu is not None:
            mu = _f64(vi_mu)
            assert mu.shape == (self.M, self.P, self.N)
        tau = None if tau is None else _f64(tau)
        obj = C.c_double()
        self._check(self.lib.vilma_set_state(self.ctx, self._stream_handle,
                                             None if mu is None else _ptr(mu), _ptr(hyper),
                                             None if tau is None else _ptr(tau), C.byref(obj)))
        return obj.value

    def initialize(self, fake_mu):
        """_initialize from the jittered ridge start [P,N] on the device; returns the ELBO of the
        starting point (hyper_delta: get_hyper())."""
        fake_mu = _f64(fake_mu)
        assert fake_mu.shape == (self.P, self.N)
        obj = C.c_double()
        self._check(self.lib.vilma_initialize(self.ctx, self._stream_handle, _ptr(fake_mu),
                                              C.byref(obj)))
        return obj.value

    def get_hyper(self):
        out = np.empty((self.A, self.M))
        self._check(self.lib.vilma_get_state(self.ctx, None, None, _ptr(out), None))
        return out

    def get_tau(self):
        out = np.empty(self.P)
        self._check(self.lib.vilma_get_state(self.ctx, None, None, None, _ptr(out)))
        return out

    def elbo(self):
        obj = C.c_double()
        self._check(self.lib.vilma_elbo(self.ctx, C.byref(obj)))
        return obj.value

    def posterior(self):
        mean, var = np.empty((self.P, self.N)), np.empty((self.P, self.N))
        self._check(self.lib.vilma_posterior(self.ctx, _ptr(mean), _ptr(var)))
        return mean, var

    def sweep(self, L, elbo, running, line_search_rate=2., flags=0):
        """One outer iteration inside the library (vilma_sweep).  L: float64[5], updated in
        place; running None = first sweep.  Returns (elbo, running, stats)."""
        assert L.dtype == np.float64 and L.shape == (5,) and L.flags.c_contiguous
        box = self._sweep_box
        if box is None:         # argument boxes are reused: this call sits in the fit's inner loop
            box = self._sweep_box = (C.c_double(), C.c_double(), _lib.SweepStats())
            self._sweep_refs = (C.byref(box[0]), C.byref(box[1]), C.byref(box[2]))
        e, r, stats = box
        e.value = elbo
        r.value = float('nan') if running is None else running
        if self.lib.vilma_sweep(self.ctx, self._stream_handle, L.ctypes.data, self._sweep_refs[0],
                                self._sweep_refs[1], line_search_rate, flags, self._sweep_refs[2]):
            self._check(1)
        return e.value, r.value, stats

    def drain(self):
        self._check(self.lib.vilma_sweep_drain(self.ctx))

    def state_form(self):
        """0: the queued sweeps held the state as a stored vi_mu; 1: as (stored vi_mu, a, c); 2: the same
        with a == 0 (no per-SNP pass reads vi_mu)."""
        f = C.c_int()
        self._check(self.lib.vilma_prof_state_form(self.ctx, C.byref(f)))
        return f.value

    def update_beta(self, L0, line_search_rate):
        """_update_beta from the current state; returns (L0 after backtracking, orig, new)."""
        L, o, n = C.c_double(float(L0)), C.c_double(), C.c_double()
        self._check(self.lib.vilma_update_beta(self.ctx, self._stream_handle, C.byref(L),
                                               float(line_search_rate), C.byref(o), C.byref(n)))
        return L.value, o.value, n.value

    def update_hyper_delta(self):
        o, n = C.c_double(), C.c_double()
        self._check(self.lib.vilma_update_hyper_delta(self.ctx, self._stream_handle, C.byref(o),
                                                      C.byref(n)))
        return o.value, n.value

    def update_error_scaling(self):
        o, n = C.c_double(), C.c_double()
        self._check(self.lib.vilma_update_error_scaling(self.ctx, self._stream_handle,
                                                        C.byref(o), C.byref(n)))
        return o.value, n.value

    # ------------------------------------------------------------------ measurement
    def prof_enable(self, on=True, every=1):
        """Bracket every `every`-th LD launch with HIP events (each pair costs a few microseconds
        of stream time, which matters on small shards)."""
        self._check(self.lib.vilma_prof_enable(self.ctx, max(1, int(every)) if on else 0))

    PROF_KINDS = ('ld_sym_kernel', 'ld_eig_fused_kernel', 'ld_sym_kernel_two_rhs',
                  'snp_pass_eval', 'snp_pass_trial', 'snp_pass_trial2',
                  'sums_pass', 'sums_pass_store', 'snp_pass_trial_lazy', 'snp_pass_trial2_lazy')

    def prof_read(self, reset=True):
        """{kernel: (milliseconds, launches)} accumulated by the library's HIP events."""
        ms = (C.c_double * len(self.PROF_KINDS))()
        n = (C.c_int64 * len(self.PROF_KINDS))()
        self._check(self.lib.vilma_prof_read(self.ctx, ms, n, 1 if reset else 0))
        return {k: (ms[i], n[i]) for i, k in enumerate(self.PROF_KINDS)}

    def stream_store(self, passes=5):
        """(milliseconds, bytes) of one bare read of the LD store as it sits in HBM: the yardstick
        for the LD kernels in this process (include/vilma_hip.h: vilma_prof_stream_store)."""
        ms, nbytes = C.c_double(), C.c_int64()
        self._check(self.lib.vilma_prof_stream_store(self.ctx, self._stream(), int(passes),
                                                     C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def stream_pattern(self, chunk_kb, scattered, grid=4096, passes=5, writes=0):
        """(milliseconds, bytes) of a bare read of the LD store in chunks of `chunk_kb`, one
        workgroup per chunk, in store order or scattered (vilma_prof_stream_pattern)."""
        ms, nbytes = C.c_double(), C.c_int64()
        self._check(self.lib.vilma_prof_stream_pattern(self.ctx, self._stream(), int(passes),
                                                       int(chunk_kb), 1 if scattered else 0,
                                                       int(grid), int(writes), C.byref(ms),
                                                       C.byref(nbytes)))
        return ms.value, nbytes.value

    def ld_order(self, order):
        """Order of ld_sym_kernel's work items (vilma_prof_ld_order); results do not change."""
        self._check(self.lib.vilma_prof_ld_order(self.ctx, int(order)))

    def ld_trace(self, buf):
        """Per-workgroup trace of ld_sym_kernel into the torch tensor buf [rows, 4] (None = off);
        only in builds with -DLD_TRACE=1 (vilma_prof_ld_trace)."""
        if buf is None:
            self._check(self.lib.vilma_prof_ld_trace(self.ctx, None, 0))
        else:
            self._check(self.lib.vilma_prof_ld_trace(self.ctx, C.c_void_p(buf.data_ptr()),
                                                     int(buf.shape[0])))
