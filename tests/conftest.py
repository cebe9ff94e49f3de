import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Build the native pieces (no-op when up to date): host library, HIP library (hipcc cross-compiles
    gfx950 without a GPU), the cbc program, and the oracle."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cbc_amd", "csrc"), "all"], stdout=subprocess.DEVNULL)
    # the lock-step emulation too: worker processes of the gloo tests would otherwise each start their own `make` of it
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu"), "libcbc_emu.so"], stdout=subprocess.DEVNULL)
    from oracle import oracle
    oracle.build()
    return True
