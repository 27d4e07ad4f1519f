"""GPU parity: lvx_ndt_align / lvx_ndt_fitness / lvx_voxel_lookup_rel (through the C ABI) against the oracle's restatement of ndt_omp's registration loop
(oracle/ndt_align.py) on the reference's own demo: the two scans of src/ndt_omp/data prepared as apps/align.cpp does (0.1 m VoxelGrid, resolution 1.0, identity
guess).  tests/test_ndt_align_oracle.py holds that oracle against the fitness scores the reference publishes (README.md:8-41).

Bars: ids exact; the fitness of a given transform bit for bit (same float distances, summed in the same order); the alignment loop — same iteration and evaluation
counts, the 6-vector to 1e-7 (the per-point arithmetic is the same float arithmetic, the sums over the points run in another order and expf / exp are the device's),
the fitness of the final transform to 1e-5 relative.
"""
import os

import numpy as np
import pytest

import lvx
from oracle import ndt_align as NA
from oracle import oracle as O

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
    return dict(tgt=tgt, src=src, td=O.voxelgrid_xyzi(tgt, 0.1), sd=O.voxelgrid_xyzi(src, 0.1))


def test_generic_neighbourhood_lookup_on_the_real_scans(ctx, clouds):
    """getNeighborhoodAtPoint(relative_coordinates, ...) (voxel_grid_covariance_omp_impl.hpp:378-408): the 26-cell table of DIRECT26, the 7-cell table, one cell, and an
    arbitrary table with a repeated and a far displacement — exact ids against the oracle, at the calibration's 0.5 m grid of a raw scan and at the demo's 1.0 m grid."""
    r26 = lvx.neighbor_cells_26()
    assert np.array_equal(r26, NA.neighbor_cells_26())
    odd = np.array([[0, 0, 0], [2, -1, 0], [0, 0, 0], [-3, 4, 1], [0, 0, -2]], np.int32)
    for cloud, leaf, q in ((clouds["tgt"], 0.5, clouds["src"]), (clouds["td"], 1.0, clouds["sd"])):
        vo = O.voxel_build(cloud, leaf)
        lvx.voxel_build(ctx, cloud, leaf, fetch=False)
        for rel in (r26, NA.REL7, NA.REL7[:1], odd):
            want = NA.voxel_lookup_rel(vo, q, np.float32(leaf), rel)
            got = lvx.voxel_lookup_rel(ctx, q, rel)
            assert np.array_equal(got, want)
        assert np.array_equal(lvx.voxel_lookup_rel(ctx, q, NA.REL7), lvx.voxel_lookup7(ctx, q))
        assert (NA.voxel_lookup_rel(vo, q, np.float32(leaf), r26) >= 0).sum() > len(q)
    assert lvx.voxel_lookup_rel(ctx, clouds["sd"][:0], r26).shape == (0, 26)


def test_fitness_score_bit_for_bit(ctx, clouds):
    td, sd = clouds["td"], clouds["sd"]
    for p in (np.zeros(6), np.array([0.4977, 0.1101, -0.0269, 0.0067, -0.0013, -0.0116])):
        M = NA.ndt_matrix(p)
        assert lvx.ndt_fitness(ctx, sd, M, td) == NA.fitness(sd, M, td)
    M = NA.ndt_matrix(np.array([0.4977, 0.1101, -0.0269, 0.0067, -0.0013, -0.0116]))
    assert lvx.ndt_fitness(ctx, sd, M, td, max_range=0.01) == NA.fitness(sd, M, td, max_range=0.01)     # getFitnessScore(max_range): only distances <= max_range count
    assert lvx.ndt_fitness(ctx, sd[:777], M, td[:1500]) == NA.fitness(sd[:777], M, td[:1500])           # ragged sizes (partial tiles on both sides)
    assert lvx.ndt_fitness(ctx, sd[:5], M, td[:0]) == np.finfo(np.float64).max                          # nothing counted


@pytest.mark.parametrize("search", [NA.DIRECT7, NA.DIRECT1, NA.DIRECT26])
def test_align_follows_the_oracle_loop(ctx, clouds, search):
    td, sd = clouds["td"], clouds["sd"]
    a = NA.NdtAligner(td, 1.0, search)
    a.align(sd)
    lvx.voxel_build(ctx, td, 1.0, fetch=False)
    r = lvx.ndt_align(ctx, sd, search=search, want_aligned=True)
    print("search %d: oracle p %s iterations %d evaluations %d | gpu p %s iterations %d evaluations %d" % (search, a.p, a.nr_iterations, a.n_eval, r["p"], r["iterations"], r["n_evaluations"]))
    assert r["iterations"] == a.nr_iterations and r["n_evaluations"] == a.n_eval and r["converged"]
    assert np.abs(r["p"] - a.p).max() <= 1e-7
    assert np.abs(r["final_transformation"] - a.final_transformation).max() <= 2e-7
    assert abs(r["trans_probability"] - a.trans_probability) <= 1e-6 * abs(a.trans_probability)
    fo = a.fitness()
    fg = lvx.ndt_fitness(ctx, sd, r["final_transformation"], td)
    print("           fitness oracle %.6f gpu %.6f" % (fo, fg))
    assert abs(fg - fo) <= 1e-5 * fo
    assert np.abs(r["aligned"][:, :3] - NA.transform_cloud(sd, r["final_transformation"])[:, :3]).max() == 0.0 and np.array_equal(r["aligned"][:, 3], sd[:, 3])


def test_align_with_line_search_iterations_and_a_guess(ctx, clouds):
    """A tight transformation_epsilon drives the loop through More-Thuente iterations and the Hessian-only pass (computeHessian, :927-928); a non-identity guess goes through
    eulerAngles (:95-111)."""
    td, sd = clouds["td"], clouds["sd"]
    lvx.voxel_build(ctx, td, 1.0, fetch=False)
    a = NA.NdtAligner(td, 1.0, NA.DIRECT7, transformation_epsilon=0.01)
    a.align(sd)
    assert any(t["mt"] > 0 for t in a.trace)
    r = lvx.ndt_align(ctx, sd, search=7, transformation_epsilon=0.01)
    assert r["iterations"] == a.nr_iterations and r["n_evaluations"] == a.n_eval
    assert np.abs(r["p"] - a.p).max() <= 1e-6
    guess = NA.ndt_matrix(np.array([0.3, 0.05, -0.02, 0.004, -0.002, -0.008]))
    a = NA.NdtAligner(td, 1.0, NA.DIRECT7)
    a.align(sd, guess=guess)
    r = lvx.ndt_align(ctx, sd, guess=guess, search=7)
    assert r["iterations"] == a.nr_iterations and np.abs(r["p"] - a.p).max() <= 1e-6
    again = lvx.ndt_align(ctx, sd, guess=guess, search=7)
    assert np.array_equal(again["p"], r["p"]) and again["score"] == r["score"]         # fixed-order reductions: run-to-run identical


def test_align_argument_errors(ctx, clouds):
    lvx.voxel_build(ctx, clouds["td"], 1.0, fetch=False)
    with pytest.raises(lvx.LvxError):
        lvx.ndt_align(ctx, clouds["sd"], search=5)
    r = lvx.ndt_align(ctx, clouds["sd"][:0])
    assert r["iterations"] == 0 and np.array_equal(r["p"], np.zeros(6))
