import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _built_library():
    """The C-ABI library is a build artefact (git-ignored): build it in-tree when a test session
    starts on a machine that has hipcc and no up-to-date copy (cross-compiles without a GPU)."""
    try:
        from vilma_amd import build
        if build.needs_build():
            build.build_library(verbose=False)
    except Exception as exc:      # no hipcc here: tests that need the library will say so
        print('libvilma_hip.so not (re)built: %r' % (exc,))
    yield
