"""CPU: the oracle's restatement of ndt_omp's registration loop (oracle/ndt_align.py + oracle/orc_ndt.cpp) on the reference's own demo — the two scans of
src/ndt_omp/data through apps/align.cpp's settings (0.1 m VoxelGrid, resolution 1.0, identity guess) — against the ONLY outputs the reference publishes about itself:
the fitness scores of src/ndt_omp/README.md:8-41 (pcl / KDTREE 0.213937, DIRECT7 0.214205, DIRECT1 0.208511).

What is held, and why not to six digits (measured in this file, numbers in DESIGN.md "NDT pin"):
  * The restated loop with the reference's defaults stops after 4 (DIRECT7) / 3 (DIRECT1) / 4 (KDTREE = pcl's own search) Newton iterations at fitness 0.204505 /
    0.224965 / 0.206161: 4.5 % / 7.9 % / 3.6 % from the README.
  * Driven to its fixed point (transformation_epsilon 1e-5) the same loop ends at 0.216489 / 0.216469 for BOTH searches: 1.1 % from the README's DIRECT7 / KDTREE
    values.  The README's three values lie between the early stop and the fixed point.
  * The statistic itself is steep: around the optimum it moves 0.5 - 0.9 % per milliradian of rotation and 0.15 % per centimetre (test below), and the default loop stops
    as soon as one clamped step is shorter than 0.1 (ndt_omp_impl.hpp:158-162) — the stop is 0.12 m short of the optimum here, on a path where the fitness swings
    between 0.204 and 0.223 from one iteration to the next.  Which iteration the published run stopped at depends on its build (PCL / Eigen versions are not pinned,
    SURVEY 8c); a different stopping iteration moves the number by several per cent, the arithmetic inside an iteration (float or double Hessian) does not move it at all.
So: order of magnitude and data preparation are confirmed by a reference-held number (the fitness is 0.075 without the 0.1 m VoxelGrid, 0.37 at 0.2 m), the converged
pose reproduces the published fitness to 1 %, the default-stop values are pinned as this oracle's own (regression), and the HIP path is held against this oracle
(tests/test_gpu_ndt_align.py).
"""
import os

import numpy as np
import pytest

from oracle import ndt_align as NA
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
README = {"pcl": 0.213937, NA.KDTREE: 0.213937, NA.DIRECT7: 0.214205, NA.DIRECT1: 0.208511}     # /root/reference/src/ndt_omp/README.md:11,16,21,26 (and :31,36,41)


@pytest.fixture(scope="module")
def clouds():
    tgt = np.load(os.path.join(GOLD, "ndt_data_251370668.npz"))["xyzi"]
    src = np.load(os.path.join(GOLD, "ndt_data_251371071.npz"))["xyzi"]
    return O.voxelgrid_xyzi(tgt, 0.1), O.voxelgrid_xyzi(src, 0.1)          # align.cpp:60-69


def test_more_thuente_pieces_known_answers():
    """updateIntervalMT / trialValueSelectionMT on hand-computed cases (ndt_omp_impl.hpp:648-768)."""
    I = [0.0, 0.0, -1.0, 0.0, 0.0, -1.0]
    assert NA._update_interval(I, 0.5, 0.2, 0.3) is False and I[3:] == [0.5, 0.2, 0.3]              # U1: f_t > f_l -> upper end
    I = [0.0, 0.0, -1.0, 0.0, 0.0, -1.0]
    assert NA._update_interval(I, 0.5, -0.2, -0.3) is False and I[:3] == [0.5, -0.2, -0.3]          # U2: lower, still descending
    I = [0.0, 0.0, -1.0, 0.0, 0.0, -1.0]
    assert NA._update_interval(I, 0.5, -0.2, 0.3) is False and I == [0.5, -0.2, 0.3, 0.0, 0.0, -1.0]   # U3: lower, ascending -> ends swap
    I = [0.5, -0.2, 0.0, 0.0, 0.0, -1.0]
    assert NA._update_interval(I, 0.5, -0.2, 0.0) is True                                            # g_t = 0: converged
    # case 1 on phi(a) = (a - 1)^2 - 1 from a_l = 0 (f 0, g -2) to a_t = 3 (f 3, g 4): cubic and quadratic minimisers are both the true minimiser 1
    assert abs(NA._trial_value(0.0, 0.0, -2.0, 0.0, 0.0, -2.0, 3.0, 3.0, 4.0) - 1.0) < 1e-12
    # case 2 (f_t <= f_l, derivatives of opposite sign) on the same parabola from a_t = 1.5 (f -0.75, g 1): secant step = 1 exactly
    assert abs(NA._trial_value(0.0, 0.0, -2.0, 0.0, 0.0, -2.0, 1.5, -0.75, 1.0) - 1.0) < 1e-12
    # division by zero follows IEEE (no exception): equal abscissae give a NaN / inf trial value that the caller clamps
    v = NA._trial_value(0.1, 0.0, -2.0, 0.0, 0.0, -2.0, 0.1, 1.0, 4.0)
    assert v != v or abs(v) == float("inf") or isinstance(v, float)


def test_transform_and_euler_round_trip():
    p = np.array([0.3, -0.2, 0.05, 0.02, -0.03, 0.4])
    M = NA.ndt_matrix(p)
    cx, sx, cy, sy, cz, sz = np.cos(p[3]), np.sin(p[3]), np.cos(p[4]), np.sin(p[4]), np.cos(p[5]), np.sin(p[5])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    assert np.abs(M[:3, :3] - Rx @ Ry @ Rz).max() < 5e-7 and np.abs(M[:3, 3] - p[:3]).max() < 1e-7 and np.array_equal(M[3], [0, 0, 0, 1])
    assert np.abs(NA.euler012(M) - p[3:]).max() < 1e-6                      # eulerAngles(0, 1, 2) inverts Rx Ry Rz (ndt_omp_impl.hpp:109)
    assert np.array_equal(np.abs(NA.euler012(np.eye(4, dtype=np.float32))), np.zeros(3))
    c = np.array([[1, 2, 3, 7], [-4, 0.5, 2, 9]], np.float32)
    t = NA.transform_cloud(c, M)
    assert np.abs(t[:, :3] - (c[:, :3].astype(np.float64) @ M[:3, :3].astype(np.float64).T + M[:3, 3])).max() < 2e-6 and np.array_equal(t[:, 3], c[:, 3])


def test_neighbourhood_tables():
    r26 = NA.neighbor_cells_26()
    assert r26.shape == (26, 3) and len({tuple(r) for r in r26}) == 26 and (0, 0, 0) not in {tuple(r) for r in r26} and np.abs(r26).max() == 1
    assert np.array_equal(r26[13:], -r26[:13])
    assert np.array_equal(NA.rel_cells(NA.DIRECT7)[0], [0, 0, 0]) and len(NA.rel_cells(NA.DIRECT1)) == 1


def test_generic_lookup_agrees_with_the_seven_cell_lookup(clouds):
    td, sd = clouds
    v = O.voxel_build(td, 1.0)
    ids7 = O.voxel_lookup7(v, sd, 1.0)
    assert np.array_equal(NA.voxel_lookup_rel(v, sd, np.float32(1.0), NA.REL7), ids7)
    ids26 = NA.voxel_lookup_rel(v, sd, np.float32(1.0), NA.neighbor_cells_26())
    r26 = NA.neighbor_cells_26()
    for k, d in enumerate(NA.REL7[1:], start=1):          # the six face neighbours are six of the 26
        col = int(np.flatnonzero((r26 == d).all(axis=1))[0])
        assert np.array_equal(ids26[:, col], ids7[:, k])
    assert (ids26 >= 0).sum() > (ids7[:, 1:] >= 0).sum()


def test_double_hessian_equals_the_float_one_to_float_rounding(clouds):
    """computeHessian (double, :540-645) and the Hessian of computeDerivatives (float per point, :484-536) are the same quantity."""
    td, sd = clouds
    a = NA.NdtAligner(td, 1.0, NA.DIRECT7)
    a.src = sd
    p = np.array([0.35, -0.1, 0.02, 0.002, -0.003, 0.01])
    tr = NA.transform_cloud(sd, NA.ndt_matrix(p))
    _, _, Hf = a._derivatives(tr, p, True)
    Hd = a._hessian(tr, p)
    assert np.abs(Hf - Hd).max() <= 1e-6 * np.abs(Hd).max() and np.abs(Hd - Hd.T).max() <= 1e-12 * np.abs(Hd).max()


@pytest.mark.parametrize("search,own,iters", [(NA.DIRECT7, 0.204505, 4), (NA.DIRECT1, 0.224965, 3), (NA.KDTREE, 0.206161, 4)])
def test_demo_alignment_default_settings(clouds, search, own, iters):
    td, sd = clouds
    a = NA.NdtAligner(td, 1.0, search)
    a.align(sd)
    f = a.fitness()
    print("search %d: fitness %.6f (README %.6f, %+.2f %%), %d iterations, %d evaluations, p = %s" % (search, f, README[search], 100 * (f / README[search] - 1), a.nr_iterations, a.n_eval, a.p))
    assert a.nr_iterations == iters and abs(f - own) <= 2e-3 * own          # this oracle's own result (regression; the loop amplifies 1e-6 m input noise to 2e-4 of the fitness)
    assert abs(f - README[search]) <= 0.10 * README[search]                 # the published value: same data, same preparation, a nearby stopping point
    assert all(t["step"] <= 0.1 + 1e-15 and t["step"] >= 0.05 - 1e-15 for t in a.trace)   # every step length clamped to [epsilon / 2, step_size] (:142, 821-822)
    scores = [t["score"] for t in a.trace]
    assert all(b > a_ for a_, b in zip(scores, scores[1:]))                 # the score rises monotonically along the accepted steps


def test_fixed_point_reproduces_the_published_fitness_to_one_percent(clouds):
    """The loop driven to its fixed point (epsilon 1e-5): both searches end at the same pose and within 1.2 % of the README's KDTREE / DIRECT7 values; the fitness around it is
    as steep as the module docstring says."""
    td, sd = clouds
    f = {}
    for search in (NA.DIRECT7, NA.DIRECT1):
        a = NA.NdtAligner(td, 1.0, search, transformation_epsilon=1e-5, max_iterations=60)
        a.align(sd)
        f[search] = (a.fitness(), a.p.copy())
        print("search %d at the fixed point: fitness %.6f, p = %s, %d iterations" % (search, f[search][0], a.p, a.nr_iterations))
    assert abs(f[NA.DIRECT7][0] - f[NA.DIRECT1][0]) <= 1e-3 * f[NA.DIRECT7][0]
    assert np.abs(f[NA.DIRECT7][1] - f[NA.DIRECT1][1]).max() < 3e-3
    for key in ("pcl", NA.DIRECT7):
        assert abs(f[NA.DIRECT7][0] - README[key]) <= 0.012 * README[key]
    p = f[NA.DIRECT7][1]
    f0 = NA.fitness(sd, NA.ndt_matrix(p), td)
    q = p.copy(); q[5] += 1e-3
    steep = abs(NA.fitness(sd, NA.ndt_matrix(q), td) - f0) / f0
    assert 2e-3 < steep < 2e-2                                               # ~0.8 % per milliradian of yaw
    assert NA.fitness(sd, np.eye(4, dtype=np.float32), td) > 1.2 * f0       # identity: 0.2709


def test_fitness_depends_on_the_preparation_as_the_readme_magnitude_requires():
    """Without align.cpp's 0.1 m VoxelGrid the same statistic is 0.075, at 0.2 m it is 0.37: the README's 0.21 confirms data and preparation."""
    tgt = np.load(os.path.join(GOLD, "ndt_data_251370668.npz"))["xyzi"]
    src = np.load(os.path.join(GOLD, "ndt_data_251371071.npz"))["xyzi"]
    p = np.array([0.4977, 0.1101, -0.0269, 0.0067, -0.0013, -0.0116])      # the fixed point of the test above
    M = NA.ndt_matrix(p)
    t2, s2 = O.voxelgrid_xyzi(tgt, 0.2), O.voxelgrid_xyzi(src, 0.2)
    f_02 = NA.fitness(s2, M, t2)
    f_raw = NA.fitness(src[::4], M, tgt)                                     # a quarter of the raw source against the whole raw target (the mean is what matters)
    assert f_raw < 0.12 and f_02 > 0.30
