"""A fit driven by a compiled C program (examples/host_fit.c: plain C99 against include/vilma_hip.h,
no Python, torch or HIP headers) reproduces the reference's recorded trajectories
(tests/golden/traj_*.npz, /root/reference/src/vilma/variational_inference.py:353-389): L bit-equal,
ELBO 1e-9 relative -- with sweeps decided on the device ahead of the host and without."""
import os
import subprocess

import numpy as np
import pytest

from helpers import golden, traj_blocks

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_host(tmp_path):
    exe = str(tmp_path / 'host_fit')
    lib_dir = os.path.join(ROOT, 'vilma_amd')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-O2',
                           '-I' + os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'host_fit.c'), '-o', exe,
                           '-L' + lib_dir, '-l:libvilma_hip.so', '-Wl,-rpath,' + lib_dir, '-lm'])
    return exe


def write_problem(g, path):
    """The flat file examples/host_fit.c reads (layout in its header)."""
    from oracle.ldop import EigenBlock
    P, N, M = int(g['P']), int(g['N']), len(g['covs'])
    A = g['annotations'].shape[1]
    f64 = lambda a: np.ascontiguousarray(a, dtype='<f8')
    se = f64(g['se'])
    if bool(g['scaled']):
        se = np.ones_like(se)
    missing = np.isclose(g['ld_diags'], 0)
    np.random.seed(int(g['seed']))            # the host's part of _initialize (:643-657)
    fake = np.random.normal(loc=np.copy(g['inverse_betas']), scale=1e-3 * se, size=(P, N))
    fake[missing] = np.nan
    n_obs = (~missing).sum(axis=0)
    col = np.where(n_obs > 0, np.nansum(fake, axis=0) / np.maximum(n_obs, 1), np.nan)
    fake[missing] = np.tile(col, [P, 1])[missing]
    fake[np.isnan(fake)] = 0.
    n_ld = int(N - len(g['missing']))
    with open(path, 'wb') as out:
        np.array([P, N, M, A, int(bool(g['scale_se'])), n_ld], dtype='<i8').tofile(out)
        for a in (g['adj_marginal_effects'], se, se ** -2 * g['ld_diags'], g['scalings']):
            f64(a).tofile(out)
        np.ascontiguousarray(np.where(g['annotations'])[1], dtype='<i4').tofile(out)
        f64(np.asarray(g['mixture_prec']).reshape(M, P, P)).tofile(out)
        for a in (g['log_det'], g['annotations'].sum(axis=0), g['chi_stat'], g['ld_ranks'], fake):
            f64(a).tofile(out)
        np.ascontiguousarray(g['perm'], dtype='<i8').tofile(out)
        t = float(g['ldthresh'])
        for blocks in traj_blocks(g):
            np.array([len(blocks)], dtype='<i8').tofile(out)
            for X in blocks:
                b = EigenBlock(X, t)             # the reference's thresholded factors, as a matrix
                np.array([X.shape[0]], dtype='<i8').tofile(out)
                f64((b.u * b.s) @ b.u.T).tofile(out)


@pytest.mark.parametrize('lookahead', [1, 0])
@pytest.mark.parametrize('name', ['p1_dense', 'p2_scale_se', 'p4_general'])
def test_c_host_reproduces_reference_trajectories(tmp_path, name, lookahead):
    g = golden('traj_%s.npz' % name)
    exe = build_host(tmp_path)
    prob = str(tmp_path / 'problem.bin')
    write_problem(g, prob)
    n = len(g['elbo'])
    res = subprocess.run([exe, prob, str(n), str(lookahead)], capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    init = float(lines[0].split()[2])
    assert abs(init - float(g['init_elbo'])) < 1e-9 * abs(init)
    sweeps = [ln.split() for ln in lines if ln.startswith('sweep ')]
    assert len(sweeps) == n
    for it, w in enumerate(sweeps):
        elbo = float(w[3])
        L = np.array([float(x) for x in w[5:10]])
        assert abs(elbo - g['elbo'][it]) < 1e-9 * abs(elbo), (it, elbo, g['elbo'][it])
        assert np.array_equal(L, g['L'][it]), (it, L, g['L'][it])
    final = float([ln for ln in lines if ln.startswith('final elbo')][0].split()[2])
    assert abs(final - g['elbo'][n - 1]) < 1e-9 * abs(final)
    sm = float([ln for ln in lines if ln.startswith('posterior_checksum')][0].split()[1])
    assert abs(sm - g['post_mean'][-1].sum()) < 1e-7 * np.abs(g['post_mean'][-1]).sum()
