import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): builds oracle/libdsk_oracle.so on demand."""
    so = os.path.join(ROOT, "oracle", "libdsk_oracle.so")
    src = os.path.join(ROOT, "oracle", "dsk_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libdsk_oracle.so"])
    from tests import oracle_py
    return oracle_py.Oracle(so)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
