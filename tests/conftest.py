import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """Tests use torch for device buffers; its bundled HIP runtime must be the one that initialises the device in this process — a liblvx.so
    context created first leaves torch without a GPU ("No HIP GPUs are available")."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # noqa: BLE001
        pass
    yield


@pytest.fixture(scope="session")
def host_check_lib():
    """g++ build of the device math (tests/native) for CPU-side Jacobian checks against the oracle."""
    import ctypes
    src = os.path.join(ROOT, "tests", "native", "resid_host_check.cpp")
    so = os.path.join(ROOT, "tests", "native", "libresid_host_check.so")
    deps = [src] + [os.path.join(ROOT, "lvi-exc_amd", "csrc", f) for f in ("lvx_math.h", "lvx_resid.h")] + [os.path.join(ROOT, "oracle", "orc_problem.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    import lvx
    made = []

    def make():
        c = lvx.Context(0)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
