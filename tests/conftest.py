import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/vgl_oracle.c."""
    so = os.path.join(ROOT, "oracle", "libvgl_oracle.so")
    src = os.path.join(ROOT, "oracle", "vgl_oracle.c")
    hdr = os.path.join(ROOT, "include", "vcfgl_hip.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvgl_oracle.so"])
    import oracle_lib
    return oracle_lib
