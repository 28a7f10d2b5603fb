"""The C-ABI library loads and exports every symbol include/*.h declares (no compute)."""
import os
import re

from vilma_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    declared = set()
    for name in ('vilma_hip.h', 'vilma_numerics.h'):
        header = open(os.path.join(ROOT, 'include', name)).read()
        declared |= set(re.findall(r'\b(vilma_[a-z0-9_]+)\s*\(', header))
    declared.discard('vilma_ctx')
    assert len(declared) >= 60
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), 'libvilma_hip.so does not export %s' % name
    assert declared == set(_lib.exported_symbols())
    assert lib.vilma_version().decode().startswith('vilma_hip')


def test_sizes_helpers():
    lib = _lib.load()
    assert lib.vilma_ld_dense_elems(5) == 5 * 16          # rows padded to 128 B
    assert lib.vilma_ld_lowrank_elems(5, 3) == 6 * 16 + 16     # U (pad2(5) x ld(3)) then s [ld(3)]
    # dense symmetric blocks keep the lower triangle by 128-column slabs
    assert lib.vilma_ld_dense_elems(588) == 128 * (588 + 460 + 332 + 204) + 76 * 80
    assert _lib.ntotals(2) == 8


def test_no_cpu_fallback_in_product():
    """Nothing under vilma_amd/ imports the oracle."""
    pkg = os.path.join(ROOT, 'vilma_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


def test_library_binds_to_the_hip_runtime_torch_ships():
    """Two HIP runtimes in one process cannot both own the GPU: whatever the import order, the
    library and torch must end up on ONE libamdhip64 (a process that loaded the library before
    torch used to see 'no HIP device available' on the GPU box)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from vilma_amd import _lib; _lib.load(); "
            "import torch; "
            "print(len(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l)))"
            % root)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, check=True)
    assert out.stdout.strip().splitlines()[-1] == '1', out.stdout + out.stderr


def test_headers_are_c99_and_the_c_host_example_links(tmp_path):
    """include/*.h are what a cgo / JNI / FFI stub binds: plain C99, no HIP or C++ types; and
    examples/host_fit.c (a whole fit from C) compiles and links against the library with gcc."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'h.c'
    src.write_text('#include "vilma_hip.h"\n#include "vilma_numerics.h"\nint main(void) { return 0; }\n')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror',
                           '-I' + os.path.join(root, 'include'), '-fsyntax-only', str(src)])
    lib_dir = os.path.join(root, 'vilma_amd')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-O2',
                           '-I' + os.path.join(root, 'include'),
                           os.path.join(root, 'examples', 'host_fit.c'), '-o', str(tmp_path / 'host_fit'),
                           '-L' + lib_dir, '-l:libvilma_hip.so', '-Wl,-rpath,' + lib_dir, '-lm'])


def test_a_rank_without_rccl_gets_an_error_code_not_a_crash():
    """vilma_comm_unique_id on a host where librccl cannot be loaded (VILMA_RCCL_LIB points at
    nothing) returns 1 with a message -- the outcome HipEngine.bind_comm's agreement protocol
    relies on -- instead of building a std::string from a null dlerror()."""
    import subprocess
    import sys
    code = ("import sys, ctypes as C; sys.path.insert(0, %r); from vilma_amd import _lib; "
            "lib = _lib.load(); buf = C.create_string_buffer(128); "
            "print(lib.vilma_comm_unique_id(buf), lib.vilma_last_error(None).decode())" % ROOT)
    env = dict(os.environ, VILMA_RCCL_LIB='/nonexistent/librccl.so')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True,
                         check=True)
    rc, msg = out.stdout.strip().splitlines()[-1].split(' ', 1)
    assert rc == '1' and 'cannot load librccl.so' in msg and '/nonexistent/librccl.so' in msg
    # twice in one process: the failure is not cached as a half-initialised binding
    code2 = code.replace("print(", "lib.vilma_comm_unique_id(buf); print(")
    out = subprocess.run([sys.executable, '-c', code2], env=env, capture_output=True, text=True,
                         check=True)
    assert out.stdout.strip().splitlines()[-1].startswith('1 ')
