"""CPU: lvi-exc_amd/csrc/lvx_stdsort.h — the restatement of libstdc++'s std::sort that the scan-registration kernel runs on one lane for sectors with EQUAL
curvatures (scanRegistration.cpp:327 sorts with the unstable std::sort) — against the real std::sort / std::partial_sort of this container's libstdc++
(the one the oracle is built with), element order included."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(ROOT, "tests", "native", "stdsort_check.cpp")
    so = os.path.join(ROOT, "tests", "native", "libstdsort_check.so")
    hdr = os.path.join(ROOT, "lvi-exc_amd", "csrc", "lvx_stdsort.h")
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in (src, hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("max_n", [16, 17, 40, 700, 5000])
def test_same_order_as_std_sort_with_equal_keys(lib, kind, max_n):
    assert lib.stdsort_mismatches(1 + kind, 400 if max_n <= 700 else 40, max_n, kind, 0) == 0


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_heap_sort_branch_equals_std_partial_sort(lib, kind):
    assert lib.stdsort_mismatches(11 + kind, 300, 900, kind, 1) == 0


def test_packed_elements_as_the_kernel_sorts_them(lib):
    rng = np.random.default_rng(5)
    for n in (1, 5, 16, 17, 300, 683):
        key = (rng.integers(0, max(n // 3, 2), n) * 0.125).astype(np.float32)        # many ties
        out = np.zeros(n, np.int32)
        assert lib.stdsort_packed(ctypes.c_int(n), key.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(np.sort(out), np.arange(n)) and (np.diff(key[out]) >= 0).all()
        assert not np.array_equal(out, np.argsort(key, kind="stable")) or n <= 16       # i.e. the order really is NOT the stable one (beyond pure insertion sort)
