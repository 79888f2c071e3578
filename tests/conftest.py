import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def vr():
    """The product package (directory name has a hyphen)."""
    lib_path = os.path.join(ROOT, "volume-rendering_amd", "libvr_hip.so")
    if not os.path.exists(lib_path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "volume-rendering_amd", "csrc")])
    return importlib.import_module("volume-rendering_amd")


@pytest.fixture(scope="session")
def oracle():
    from helpers import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    from helpers import Golden
    return Golden()


@pytest.fixture(scope="session")
def gpu(vr):
    """One renderer context on cuda:0 — fails (does not skip) if the HIP path is unusable."""
    import torch  # noqa: F401  (loads the process's HIP runtime first)
    r = vr.HipRenderer(0)
    yield r
    r.close()
