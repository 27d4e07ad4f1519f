"""The upstream kernels on the only REAL data the reference ships: the two LiDAR scans of ndt_omp's demo (src/ndt_omp/data/251370668.pcd = target, 251371071.pcd
= source; apps/align.cpp:25-52), committed as point arrays (tests/golden/ndt_data_*.npz, made by tests/golden/make_ndt_data.py).

Settings: align.cpp's (0.1 m pcl::VoxelGrid down-sampling of both clouds :60-69, NDT resolution 1.0 :85,96, identity initial guess) and the calibration's
(lvi.yaml:26 ndt_resolution 0.5 on the raw cloud; setSurfelMap thresholds of surfel_association.h).  Real scans are what the synthetic clouds are not:
69 k points with range-dependent density, ground rings, thin structures, leaves with 1-5 points next to leaves with hundreds.

Bars as for the synthetic clouds (tests/upstream_checks.py): leaf keys / counts / point lists and DIRECT7 / DIRECT1 ids exact, moments 1e-12, surfel leaves / boxes / inlier
counts exact, NDT score / gradient / Hessian 1e-5 (float per-point arithmetic, double accumulation in another order).
"""
import os

import numpy as np
import pytest

import lvx
from oracle import oracle as O
from upstream_checks import check_voxels

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = lvx.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def clouds():
    tgt = np.load(os.path.join(GOLD, "ndt_data_251370668.npz"))["xyzi"]
    src = np.load(os.path.join(GOLD, "ndt_data_251371071.npz"))["xyzi"]
    assert tgt.shape == (69088, 4) and src.shape == (69792, 4)
    # align.cpp:60-69 — both clouds through pcl::VoxelGrid(0.1 m) before registration (input preparation, identical for both sides)
    return dict(tgt=tgt, src=src, tgt_ds=O.voxelgrid_xyzi(tgt, 0.1), src_ds=O.voxelgrid_xyzi(src, 0.1))


@pytest.mark.parametrize("which,leaf", [("tgt_ds", 1.0), ("tgt", 1.0), ("tgt", 0.5), ("src", 0.5)])
def test_voxel_covariance_grid_of_a_real_scan(ctx, clouds, which, leaf):
    cloud = clouds[which]
    vo = O.voxel_build(cloud, leaf)
    vg = lvx.voxel_build(ctx, cloud, leaf)
    check_voxels(vg, vo)
    assert vo["n_leaves"] > 1000 and (vo["leaf_n"] >= 6).sum() > 500
    q = clouds["src_ds"] if which == "tgt_ds" else clouds["src"]          # the other scan as queries, as registration does
    ids7 = O.voxel_lookup7(vo, q, leaf)
    assert np.array_equal(lvx.voxel_lookup7(ctx, q), ids7)
    assert np.array_equal(lvx.voxel_lookup1(ctx, q), ids7[:, 0])
    assert (ids7[:, 0] >= 0).sum() > 0.4 * len(q)                           # consecutive scans of one drive overlap


def _transform(cloud, p6):
    cx, sx, cy, sy, cz, sz = np.cos(p6[3]), np.sin(p6[3]), np.cos(p6[4]), np.sin(p6[4]), np.cos(p6[5]), np.sin(p6[5])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    out = cloud.copy()
    out[:, :3] = (cloud[:, :3] @ (Rx @ Ry @ Rz).astype(np.float32).T + np.asarray(p6[:3], np.float32)).astype(np.float32)
    return out


def test_ndt_derivatives_between_the_two_real_scans(ctx, clouds):
    """computeDerivatives (DIRECT7) of align.cpp's registration problem: down-sampled source against the resolution-1.0 grid of the down-sampled target, at the
    identity guess and at a pose one Newton step could reach."""
    tgt, src = clouds["tgt_ds"], clouds["src_ds"]
    vo = O.voxel_build(tgt, 1.0)
    lvx.voxel_build(ctx, tgt, 1.0, fetch=False)
    for p6 in (np.zeros(6), np.array([0.35, -0.1, 0.02, 0.002, -0.003, 0.01])):
        tr = _transform(src, p6)
        so, go, Ho = O.ndt_derivatives(vo, 1.0, src, tr, p6)
        sg, gg, Hg = lvx.ndt_derivatives(ctx, src, tr, p6)
        assert so > 1000 and abs(sg - so) <= 1e-6 * abs(so)
        assert np.abs(gg - go).max() <= 1e-5 * np.abs(go).max() and np.abs(Hg - Ho).max() <= 1e-5 * np.abs(Ho).max()


@pytest.mark.parametrize("which", ["tgt", "src"])
def test_surfel_map_of_a_real_scan(ctx, clouds, which):
    """setSurfelMap at the calibration's settings (0.5 m grid, planarity 0.7, 0.05 m fit, >= 10 points, >= 20 inliers) on a raw real scan, then that map
    drives the association kernel with the scan itself, organised as rings of its own points."""
    cloud = clouds[which]
    vo = O.voxel_build(cloud, 0.5)
    ro = O.surfel_extract(cloud, vo)
    lvx.voxel_build(ctx, cloud, 0.5, fetch=False)
    rg, n = lvx.surfel_extract(ctx, max_planes=vo["n_leaves"])
    assert n == len(ro["leaf"]) and n > 100
    assert np.array_equal(rg["leaf"], ro["leaf"]) and np.array_equal(rg["n_points"], ro["n_points"])
    assert np.array_equal(rg["n_inliers"], ro["n_inliers"]) and np.array_equal(rg["plane_type"], ro["plane_type"])
    assert np.abs(rg["p4"] - ro["p4"]).max() <= 1e-9 and np.abs(rg["Pi"] - ro["Pi"]).max() <= 1e-9 * np.abs(ro["Pi"]).max()
    assert np.array_equal(rg["box_min"], ro["box_min"]) and np.array_equal(rg["box_max"], ro["box_max"])
    H, W = 32, len(cloud) // 32
    scan = cloud[:H * W].reshape(H, W, 4)
    fo = O.surfel_assoc(scan, ro["p4"], ro["box_min"], ro["box_max"], 0.05, 2)
    fg = lvx.surfel_assoc(ctx, scan, rg["p4"], rg["box_min"], rg["box_max"], 0.05, 2)
    assert np.array_equal(fg, fo) and (fo >= 0).sum() > 100
    # several copies per call take the grid path (more than two scans)
    scans = np.stack([scan, scan[:, ::-1], scan])
    fgb, _ = lvx.surfel_assoc_emit(ctx, scans, np.zeros((3, H, W), lvx.POINT_XYZIT), rg["p4"], rg["box_min"], rg["box_max"], 0.05, 2)
    assert np.array_equal(fgb[0], fo) and np.array_equal(fgb[2], fo)
    assert np.array_equal(fgb[1], O.surfel_assoc(np.ascontiguousarray(scan[:, ::-1]), ro["p4"], ro["box_min"], ro["box_max"], 0.05, 2))
