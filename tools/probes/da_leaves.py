"""Leaf-length statistics of the map cloud of one lvx_data_association round (what k_vx_leaf sees in the pipeline)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
torch.cuda.init()
import importlib
lvx = importlib.import_module("lvi-exc_amd.lvx"); synth = importlib.import_module("lvi-exc_amd.synth")
S = synth.make_sequence(seed=50)
g = lvx.Context(0)
g.set_spline(S["t0"], S["dt"], S["n_knots"])
rawd = np.zeros(S["scans"].shape, dtype=lvx.POINT_XYZIT)
for k in ("x", "y", "z", "timestamp"):
    rawd[k] = S["scans"][k]
g._ck(g._l.lvx_set_scans(g._h, C.c_int(len(rawd)), C.c_int(S["H"]), C.c_int(S["W"]), rawd.ctypes.data_as(C.c_void_p)))
st_ = np.ascontiguousarray(S["state0"], np.float64)
npl, npt = C.c_int32(0), C.c_int32(0)
g._ck(g._l.lvx_data_association(g._h, st_.ctypes.data_as(C.c_void_p), C.c_double(S["t_map"]), None, C.byref(npl), C.byref(npt)))
info = g.voxel_info()
nl = info["n_leaves"]; n = info["n_points"]
offsets = np.zeros(nl + 1, np.int32); keys = np.zeros(nl, np.int32)
NULL = C.c_void_p(0)
g._ck(g._l.lvx_voxel_get(g._h, keys.ctypes.data_as(C.c_void_p), NULL, NULL, NULL, NULL, NULL, NULL, NULL, offsets.ctypes.data_as(C.c_void_p), NULL))
cnt = np.diff(offsets)
print("points", n, "leaves", nl, "in leaves", offsets[-1], "mean", cnt.mean(), "pct 50/90/99/100", np.percentile(cnt, [50, 90, 99, 100]))
print("longest leaves", np.sort(cnt)[-8:], "timestamp==0 points", int((rawd["timestamp"] == 0).sum()))
