import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): builds oracle/liboracle.so on demand."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from tests.oracle_binding import Oracle
    return Oracle(so)


@pytest.fixture(scope="session")
def backend():
    """HIP backend through the C ABI; fails loudly if the .so or the GPU is missing."""
    from cairo_m_amd import Backend
    return Backend(0)
