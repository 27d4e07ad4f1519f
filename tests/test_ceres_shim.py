"""The keep-Ceres drop-in (lvi-exc_amd/host/lvx_ceres_shim.hpp): LvxEvaluationCallback + one LvxRowBlock per residual block, driven through the
ceres::EvaluationCallback / ceres::CostFunction virtual interfaces of a stand-in <ceres/ceres.h> (the image has no Ceres).
CPU: the header compiles against those interfaces and links the C ABI; PackState / UnpackState round-trip mock entities with the reference's
accessor names and land on the documented state offsets; the segment construction matches hand-checked cases of spline_base.h:380-424.
GPU: Evaluate(parameters, residuals, jacobians) of every block of the ambient fixture problem — residual rows, parameter-block order and
sizes, ambient Jacobian blocks — against tests/golden/ambient_small.npz after the manifold projection Ceres applies (J . P)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import lvx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
LIBDIR = os.path.join(ROOT, "lvi-exc_amd")


@pytest.fixture(scope="module")
def shim_lib():
    import build as lvx_build
    lvx_build.build()
    src, so = os.path.join(NATIVE, "ceres_shim_check.cpp"), os.path.join(NATIVE, "libceres_shim_check.so")
    deps = [src, os.path.join(LIBDIR, "host", "lvx_ceres_shim.hpp"), os.path.join(ROOT, "include", "lvx.h"), os.path.join(NATIVE, "mock_ceres", "ceres", "ceres.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + os.path.join(NATIVE, "mock_ceres"), "-I" + os.path.join(LIBDIR, "host"),
                               src, "-o", so, "-L" + LIBDIR, "-llvx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return C.CDLL(so)


def test_shim_compiles_and_host_pieces_work(shim_lib):
    assert shim_lib.shim_segments_check() == 0
    assert shim_lib.shim_packstate_roundtrip(7, 3) == 0
    assert shim_lib.shim_packstate_roundtrip(4, 0) == 0


@pytest.mark.gpu
def test_cost_functions_hand_out_the_ambient_blocks(shim_lib):
    import test_ambient_pin as A
    P = A._load()
    N, L = P["n_knots"], P["n_landmarks"]
    g = lvx.Context(0)
    lvx.load_problem(g, P, A.TAU)
    g.set_orientation_prior(P["prior_t"], P["prior_q_wxyz"], P["prior_w"])
    ns, nres = g.state_size, g.layout()["n_residuals"]
    res, J = np.zeros(nres), np.zeros((nres, ns))
    d = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
    i = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(C.c_void_p)
    state = np.ascontiguousarray(P["state"], np.float64)
    keep = [np.ascontiguousarray(P[k], np.float64) for k in ("t_imu", "surf_t", "rep_t0", "lm_t0")] + [np.ascontiguousarray(P[k], np.int32) for k in ("rep_lm", "cs_lm")]
    rc = shim_lib.shim_check_all(g._h, d(state), C.c_int(ns), C.c_double(P["t0"]), C.c_double(P["dt"]), C.c_int(N), C.c_int(L), C.c_double(P["camera"]["readout"]), C.c_uint(A.TAU),
                                 C.c_int(len(P["t_imu"])), keep[0].ctypes.data_as(C.c_void_p), C.c_int(1), C.c_double(P["prior_t"]), C.c_int(len(P["surf_t"])), keep[1].ctypes.data_as(C.c_void_p),
                                 C.c_double(P["t_map"]), C.c_int(len(P["rep_lm"])), keep[4].ctypes.data_as(C.c_void_p), keep[2].ctypes.data_as(C.c_void_p), keep[3].ctypes.data_as(C.c_void_p),
                                 C.c_int(len(P["cs_lm"])), keep[5].ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p), J.ctypes.data_as(C.c_void_p))
    assert rc == 0
    g.close()
    r_ref, Jt_ref, _, _, _ = A._reference_system(P, A._free(P, A.TAU))
    assert np.abs(res - r_ref).max() <= 1e-11 * max(np.abs(r_ref).max(), 100.0)
    T = A._tangent_map(P["state"], N, L)
    Jt = J @ T                                       # what Ceres forms from the blocks: J_ambient . P
    fmax = np.array([np.abs(Jt_ref[P["row_family"] == f]).max() for f in P["row_family"]])[:, None]
    scale = np.maximum(np.abs(Jt_ref).max(axis=1, keepdims=True), 1e-6 * fmax)
    locked = np.ones(Jt.shape[1], bool); locked[A._free(P, A.TAU)] = False
    assert (np.abs(Jt - Jt_ref)[:, ~locked] / scale).max() <= 1e-9
    # the blocks carry no component along q (it would be annihilated by P anyway): J_q . q = 0 for every quaternion block
    for so in [3 * N + 4 * k for k in range(N)] + [7 * N + 16, 7 * N + 24]:
        assert np.abs(J[:, so:so + 4] @ P["state"][so:so + 4]).max() <= 1e-9 * max(1.0, np.abs(J[:, so:so + 4]).max())
