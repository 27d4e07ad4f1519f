#!/usr/bin/env python
"""Regenerate tests/golden/*.npz: small seeded inputs + the CPU oracle's outputs for them.

These are REGRESSION vectors of this repository's oracle (the reference ships no golden vectors and cannot be built or run
here — DESIGN.md §4), so they pin the oracle against accidental change; they do not pin it to the reference.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd"))
import lvx  # noqa: E402
import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def solve_problem():
    P = synth.make_problem(seed=77, duration=0.6, n_surfel=120, n_planes=6, n_landmarks=8, n_camsurf=3)
    o = O.Oracle()
    lvx.load_problem(o, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    r = o.evaluate(P["state0"], normal_eq=True)
    keep = {k: P[k] for k in ("t0", "dt", "n_knots", "t_imu", "gyro", "acc", "w_gyro", "w_acc", "planes", "surf_pt", "surf_t", "surf_plane", "t_map", "huber_surf",
                              "w_surf", "n_landmarks", "lm_uv", "lm_t0", "rep_lm", "rep_uv", "rep_t0", "huber_rep", "w_rep", "cs_lm", "cs_plane", "huber_cs", "w_cs", "state0")}
    keep["camera"] = np.array([P["camera"][k] for k in ("rows", "cols", "readout", "fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "solve_small.npz"), cost=r["cost"], residuals=r["residuals"], g=r["g"], H_diag=np.diag(r["H"]).copy(),
                        H_frob=np.linalg.norm(r["H"]), **keep)


def upstream():
    pts = synth.make_vlp16_sweep(seed=1, n_az=360)
    r = O.scan_register(pts, 16, 0.3)
    np.savez_compressed(os.path.join(HERE, "scanreg_small.npz"), pts=pts, **{k: r[k] for k in ("cloud", "curvature", "label", "sort_ind", "picked", "scan_start",
                                                                                              "scan_end", "sharp", "less_sharp", "flat", "less_flat")})
    cloud = synth.make_voxel_cloud(seed=2, n=4000)
    v = O.voxel_build(cloud, 1.0)
    q = synth.rigid_move(cloud)[:500]
    np.savez_compressed(os.path.join(HERE, "voxel_small.npz"), cloud=cloud, queries=q, ids7=O.voxel_lookup7(v, q, 1.0),
                        **{k: v[k] for k in ("grid", "leaf_key", "leaf_n", "mean", "cov", "icov", "evals", "offsets", "point_ids")})
    scan, p4, bmin, bmax = synth.make_assoc_problem(seed=5, H=16, W=450, n_planes=300)
    np.savez_compressed(os.path.join(HERE, "assoc_small.npz"), scan=scan, p4=p4, bmin=bmin, bmax=bmax, flag=O.surfel_assoc(scan, p4, bmin, bmax, 0.05, 2))


if __name__ == "__main__":
    solve_problem()
    upstream()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
