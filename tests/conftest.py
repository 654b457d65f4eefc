import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def seopt():
    """Developer switches of the library for ONE test: the library reads its switches from the environment once per
    process (no getenv on the launch path), so a test that compares two kernel forms in one process changes them through
    se_debug_set_option; everything is restored afterwards."""
    from sketchedit_amd import _lib

    class _Opts:
        def __init__(self):
            self._orig = {}

        def set(self, name, value):
            self._orig.setdefault(name, _lib.get_option(name))
            _lib.set_option(name, int(value))

        def unset(self, name):
            if name in self._orig:
                _lib.set_option(name, self._orig[name])

    o = _Opts()
    yield o
    for name, v in o._orig.items():
        _lib.set_option(name, v)
