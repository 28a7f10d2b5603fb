"""Build libvilma_hip.so (the hand-written gfx950 kernels + C-ABI) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libvilma_hip.so')
SOURCES = ['kernels.hip', 'capi.hip', 'sweep.hip', 'numerics_api.hip']
HEADERS = ['kernels.h', 'ctx.h', os.path.join('..', '..', 'include', 'vilma_hip.h'),
           os.path.join('..', '..', 'include', 'vilma_numerics.h')]


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, extra_flags=(), out=None):
    """hipcc --offload-arch=gfx950 -O3 -shared; cross-compiles without a GPU.  `extra_flags` /
    `out` build a variant next to the default library (see VILMA_HIP_LIB in _lib.py)."""
    if out is None and not force and not needs_build():
        return LIB
    cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-ffp-contract=on', '-Wall', '-Wno-unused-function'] + list(extra_flags) + [
           '-o', out or LIB] + [os.path.join(CSRC, s) for s in SOURCES] + ['-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
