"""Build libvilma_hip.so (the hand-written gfx950 kernels + C-ABI) in-tree with hipcc."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libvilma_hip.so')
OBJ = os.path.join(CSRC, '_obj')        # per-source objects (git-ignored: *.o)
SOURCES = ['kernels.hip', 'capi.hip', 'sweep.hip', 'numerics_api.hip']
HEADERS = ['kernels.h', 'ctx.h', os.path.join('..', '..', 'include', 'vilma_hip.h'),
           os.path.join('..', '..', 'include', 'vilma_numerics.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-Wall',
         '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, extra_flags=(), out=None):
    """hipcc --offload-arch=gfx950 -O3: one object per source (compiled side by side, recompiled
    only when the source or a header is newer), then one -shared link; cross-compiles without a
    GPU.  `extra_flags` / `out` build a variant next to the default library (see VILMA_HIP_LIB in
    _lib.py) with objects of its own."""
    if out is None and not force and not needs_build():
        return LIB
    variant = out is not None or bool(extra_flags)
    tag = ('_' + os.path.splitext(os.path.basename(out))[0]) if out else ('_flags' if extra_flags else '')
    objdir = OBJ + tag
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _newest_header()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + '.o')
        stale = (force or variant or not os.path.exists(obj)
                 or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t))
        jobs.append((src, obj, stale))

    def compile_one(job):
        src, obj, stale = job
        if not stale:
            return
        cmd = [_hipcc()] + FLAGS + list(extra_flags) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
        list(pool.map(compile_one, jobs))
    link = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out or LIB] + \
           [obj for _, obj, _ in jobs] + ['-ldl']
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    return out or LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
