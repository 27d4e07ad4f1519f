"""The 16 x 16 building block of the register-resident batched Cholesky (lvi-exc_amd/csrc/lvx_chol16.h: outer-product factorisation with one FP64 MFMA per column, the inverse of the
triangle carried in the same accumulator tile), run on its own by tools/probes/chol16_probe (built by __graft_entry__.build(), or here with hipcc if it is missing):
U^T U = A and inv(L) L = I to rounding on random, scaled and weakly definite matrices, exact zeros above the diagonal of the inverse, the first non-positive pivot reported."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "probes", "chol16_probe")


def _lines():
    if not os.path.exists(EXE):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", "-I", os.path.join(ROOT, "lvi-exc_amd", "csrc"),
                               EXE + ".hip", "-o", EXE])
    out = subprocess.run([EXE, "check"], capture_output=True, text=True, timeout=120, check=True).stdout
    rows = [ln.split() for ln in out.splitlines() if ln.startswith("CHECK")]
    return {r[1]: (float(r[2]), float(r[3]), int(r[4]), float(r[5])) for r in rows}


def test_factor_and_inverse_of_a_16x16_tile():
    res = _lines()
    assert len(res) >= 14
    for name, (e_fact, e_inv, bad, upper) in res.items():
        if name.startswith("negative_pivot"):
            assert bad == 6                      # 1-based first non-positive pivot; the pivot is replaced by 1 and nothing turns into NaN
            assert e_fact == e_fact and e_inv == e_inv
            continue
        assert bad == 0 and upper == 0.0
        assert e_fact <= 2e-15                   # |U^T U - A| / max |A|
        assert e_inv <= (1e-13 if name == "weak" else 3e-15)
