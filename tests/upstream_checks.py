"""Comparison helpers shared by the upstream-kernel parity tests (GPU result dict vs oracle result dict)."""
import numpy as np


def check_voxels(vg, vo):
    """voxel grid: leaf keys / counts / point lists exact; mean, cov 1e-12 rel; eigenvalues 1e-10 rel; eigenvectors up to sign; inverse covariance 1e-8 rel"""
    assert vg["n_leaves"] == vo["n_leaves"] and np.array_equal(vg["grid"], vo["grid"])
    assert np.array_equal(vg["leaf_key"], vo["leaf_key"]) and np.array_equal(vg["leaf_n"], vo["leaf_n"])
    assert np.array_equal(vg["offsets"], vo["offsets"]) and np.array_equal(vg["point_ids"][:vo["offsets"][-1]], vo["point_ids"][:vo["offsets"][-1]])
    assert np.allclose(vg["mean"], vo["mean"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(vg["centroid"].view(np.uint32), vo["centroid"].view(np.uint32))
    ok = vo["leaf_n"] >= 6
    sc = np.abs(vo["cov"][ok]).max(axis=1, keepdims=True)
    assert (np.abs(vg["cov"][ok] - vo["cov"][ok]) <= 1e-12 * sc + 1e-18).all()
    assert np.allclose(vg["evals"][ok], vo["evals"][ok], rtol=1e-10, atol=1e-16)
    si = np.abs(vo["icov"][ok]).max(axis=1, keepdims=True)
    assert (np.abs(vg["icov"][ok] - vo["icov"][ok]) <= 1e-8 * si).all()
    Vg, Vo = vg["evecs"][ok].reshape(-1, 3, 3), vo["evecs"][ok].reshape(-1, 3, 3)
    ev = vo["evals"][ok]
    sep = (np.diff(ev, axis=1).min(axis=1) > 1e-6 * ev[:, 2])       # eigenvectors only comparable for separated eigenvalues
    dots = np.abs(np.einsum("nij,nij->nj", Vg[sep], Vo[sep]))
    assert (dots > 1 - 1e-8).all()
