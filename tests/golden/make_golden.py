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


def next_rows():
    """Rows added after the first fixtures: scan de-skew, surfel map extraction, free time-offset Jacobian columns."""
    P = synth.make_problem(seed=78, duration=0.8, n_surfel=60, n_planes=4, n_landmarks=6, n_camsurf=2)
    o = O.Oracle()
    lvx.load_problem(o, P, 0)                               # both sensor time offsets free
    N = P["n_knots"]
    s = P["state0"].copy(); s[7 * N + 23] = 2e-4; s[7 * N + 31] = -3e-4
    r = o.evaluate(s, jac=True, normal_eq=True)
    J = O.dense_jacobian(r["jac_cols"], r["jac_vals"], o.tangent_size)
    keep = {k: P[k] for k in ("t0", "dt", "n_knots", "t_imu", "gyro", "acc", "w_gyro", "w_acc", "planes", "surf_pt", "surf_t", "surf_plane", "t_map", "huber_surf",
                              "w_surf", "n_landmarks", "lm_uv", "lm_t0", "rep_lm", "rep_uv", "rep_t0", "huber_rep", "w_rep", "cs_lm", "cs_plane", "huber_cs", "w_cs")}
    keep["camera"] = np.array([P["camera"][k] for k in ("rows", "cols", "readout", "fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")], dtype=np.float64)
    rng = np.random.default_rng(5)
    raw = np.zeros(400, dtype=lvx.POINT_XYZIT)
    raw["x"], raw["y"], raw["z"] = rng.uniform(-15, 15, (3, 400)).astype(np.float32)
    raw["intensity"] = rng.uniform(0, 200, 400).astype(np.float32)
    raw["timestamp"] = rng.uniform(P["t_start"], P["t_end"], 400)
    raw["x"][::41] = np.nan
    raw["timestamp"][7::53] = P["t0"] - 1.0
    st = P["state_true"]
    q0, p0, _ = O.eval_lidar_pose(o, st, [P["t_map"]])
    und = O.undistort(o, st, raw, synth.qconj(q0[0]), p0[0], True)
    np.savez_compressed(os.path.join(HERE, "tau_deskew_small.npz"), state=s, cost=r["cost"], residuals=r["residuals"], J_tau_lidar=J[:, 6 * N + 14].copy(),
                        J_tau_cam=J[:, 6 * N + 21].copy(), g=r["g"], raw=raw.view(np.uint8), state_true=st, q_map=q0[0], p_map=p0[0], undistorted=und, **keep)
    cloud = synth.make_voxel_cloud(seed=3, n=20000)
    v = O.voxel_build(cloud, 0.5)
    e = O.surfel_extract(cloud, v)
    np.savez_compressed(os.path.join(HERE, "surfel_extract_small.npz"), cloud=cloud, **e)


if __name__ == "__main__":
    solve_problem()
    upstream()
    next_rows()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
