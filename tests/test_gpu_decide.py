"""Decisions on the device: vilma_decide reproduces the host's accept test bit for bit, work
queued under a flag that turned out 0 leaves every buffer untouched, and the two-phase fetch
returns what the blocking fetch returns."""
import math

import numpy as np
import pytest

from helpers import golden, product_vi_from_traj

pytestmark = pytest.mark.gpu


def _ready(name, sweeps=2):
    from vilma_amd.variational_inference import REL_TOL, ABS_TOL
    g = golden('traj_%s.npz' % name)
    vi, _ = product_vi_from_traj(g)
    np.random.seed(int(g['seed']))
    params = vi._initialize()
    elbo, L, red = vi.elbo(params), np.ones(5), None
    for _ in range(sweeps):                 # with --learn-scaling tau moves away from 1
        params, L, elbo, red = vi._optimize_step(params, L, elbo, 2., red)
    vi.engine.drain()
    half = np.array([0.5 * vi.ld_ranks[p] * math.log(vi.error_scaling[p])
                     for p in range(vi.num_pops)])
    return vi, half, REL_TOL, ABS_TOL


@pytest.mark.parametrize('name', ['p2_scale_se', 'p1_dense', 'p4_general'])
def test_decide_matches_host_bitwise(name):
    vi, half, rel, ab = _ready(name)
    eng, Lay = vi.engine, vi.engine.layout
    eng.eval()
    eng.accept(False)
    seen = set()
    for step in (1.0, 0.5, 0.03, 4.0, 64.0):           # large steps overshoot: rejected
        eng.trial(step)
        eng.decide(vi.chi_stat, half, rel, ab, False, 0)
        flag, (orig, new) = eng.read_decision(0)
        host = eng.fetch()
        want_orig = vi._objective_from(host[Lay.totals])
        want_new = vi._objective_from(host[Lay.ttotals])
        assert orig == want_orig and (new == want_new or (np.isnan(new) and np.isnan(want_new)))
        assert flag == int(want_new >= want_orig - rel * abs(want_orig) - ab)
        seen.add(flag)
    assert seen == {0, 1}                               # both outcomes were exercised
    # the convergence veto: dsum[0] == 0 after two identical snapshots
    eng.snapshot_mean(); eng.mean_diff()
    eng.trial(0.03)                                     # accepted above
    eng.decide(vi.chi_stat, half, rel, ab, True, 1)
    assert eng.read_decision(1)[0] == 0
    eng.decide(vi.chi_stat, half, rel, ab, False, 1)
    assert eng.read_decision(1)[0] == 1
    eng.close()


def test_predicated_work_is_skipped_when_the_flag_is_zero():
    vi, half, rel, ab = _ready('p2_scale_se')
    eng = vi.engine
    eng.eval()
    eng.accept(False)
    eng.delta_sums()
    before = eng.fetch()
    mu0, (m0, v0) = eng.get_mu(), eng.get_moments()
    eng.trial(0.03)
    eng.decide(vi.chi_stat, half, rel, -1e300, False, 0)       # impossible bar: flag 0
    assert eng.read_decision(0)[0] == 0
    trial_state = eng.fetch()
    eng.spec_save()
    eng.set_predicate(0)                                        # a whole speculative stage
    eng.accept(True); eng.mstep(); eng.eval(diff=True); eng.accept(False)
    eng.trial(0.5); eng.delta_sums(1)
    eng.decide(vi.chi_stat, half, rel, ab, False, 1, snapshot=True)
    eng.set_predicate(None)
    eng.fetch_begin(1, snapshot=True)
    _, flags = eng.fetch_end(1)          # a dead stage reports its flags; its snapshot is stale
    after = eng.fetch()
    eng.spec_restore()
    assert flags == (0, 0)                                      # the chained decision is 0 too
    assert np.array_equal(after, trial_state)                   # nothing wrote the result vector
    assert np.array_equal(eng.get_mu(), mu0)
    m1, v1 = eng.get_moments()
    assert np.array_equal(m1, m0) and np.array_equal(v1, v0)
    # and the same stage under a flag that is 1 equals the unpredicated sequence
    eng.trial(0.03)
    eng.decide(vi.chi_stat, half, rel, 1e300, False, 0)         # flag 1
    eng.set_predicate(0)
    eng.accept(True); eng.mstep(); eng.eval(); eng.accept(False)
    eng.set_predicate(None)
    got = eng.fetch()
    assert eng.read_decision(0)[0] == 1
    assert not np.array_equal(got[eng.layout.totals], before[eng.layout.totals])
    assert np.isfinite(vi._objective_from(got[eng.layout.totals]))
    eng.close()


def test_two_phase_fetch_equals_fetch():
    vi, half, rel, ab = _ready('p1_dense', sweeps=1)
    eng = vi.engine
    eng.eval()
    eng.fetch_begin(0)
    a, _ = eng.fetch_end(0)
    assert np.array_equal(a, eng.fetch())
    # the snapshot taken by the decide kernel, copied out on the copy stream, is the same vector
    # even though the result vector is overwritten right behind it
    eng.accept(False)
    eng.trial(0.5)
    want = eng.fetch()
    eng.decide(vi.chi_stat, half, rel, ab, False, 1, snapshot=True)
    eng.fetch_begin(1, snapshot=True)
    eng.eval()                                  # overwrites the totals slice of the live vector
    b, flags = eng.fetch_end(1)
    assert np.array_equal(b, want) and flags[1] in (0, 1)
    eng.close()
