import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import cases
    return cases.load_golden()


EXP_LIB = os.path.join(ROOT, "diffassemble_amd", "lib_exp", "libdiffassemble_hip.so")


def exp_env(**extra):
    """Environment of a subprocess that runs on the EXPERIMENTS build of the library (DA_EXPERIMENTS=1 python __graft_entry__.py -> lib_exp/):
    the only build that holds the A/B variants that lost and reads their DA_* switches.  Skips the calling test when that build is absent."""
    if not os.path.exists(EXP_LIB):
        pytest.skip("experiments build absent (DA_EXPERIMENTS=1 python __graft_entry__.py)")
    return dict(os.environ, DA_LIB_PATH=EXP_LIB, **extra)
