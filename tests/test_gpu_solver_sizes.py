"""The block-cyclic-reduction solver picks its kernels by the block size b = half-bandwidth of the band (reprojection blocks couple the knots of a landmark's two views):
register-resident Cholesky / LDS-resident forward solves / own Schur-update kernel in their 8-, 12- and 13-tile instances (b <= 128, 192, 208), the streaming triangular solve
where the factor does not fit the LDS (b > 192), rocSOLVER + rocBLAS beyond 208.  Config 4 (b = 180) has its own tests (test_gpu_fullsize_oracle.py); here every other
instance solves a problem of its size and is held against the sequential band Cholesky (LVX_SOLVER_SEQ=1: an independent elimination order) and against the LM system itself."""
import numpy as np
import pytest

import lvx
import synth

pytestmark = pytest.mark.gpu
LOCKS = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
RADIUS = 1e4


# (views per landmark, camera rate) -> block size: 4-tile, 8-tile, 12-tile (not config 4's 180), 12-tile at the LDS limit of the forward solves (8 instead of 12 wavefronts),
# 13-tile (streaming forward solves), beyond the own kernels (rocSOLVER / rocBLAS / streaming solves without the diagonal-triangle inverses)
@pytest.mark.parametrize("views,cam_rate,lo_hi", [(4, 20.0, (16, 64)), (6, 20.0, (65, 128)), (8, 20.0, (129, 176)), (11, 20.0, (192, 192)), (11, 19.0, (193, 208)), (13, 20.0, (209, 256))],
                         ids=lambda v: str(v))
def test_step_of_every_kernel_instance_matches_the_sequential_band_cholesky(views, cam_rate, lo_hi):
    P = synth.make_bench_problem(seed=21 + views, n_imu=6000, n_surfel=30000, n_reproj=2400, n_planes=60, views_per_lm=views, cam_rate=cam_rate)
    g = lvx.Context(0)
    lvx.load_problem(g, P, LOCKS)
    x = P["state0"]
    g.evaluate(x, normal_eq=True, dense=False, residuals=False)
    lo = g.layout()
    b = 4 * ((lo["bandwidth"] + 3) // 4)
    assert lo_hi[0] <= b <= lo_hi[1], "the generator no longer lands in this kernel instance's range: b = %d" % b
    d, m = g.solve_step(RADIUS, True)
    assert g.layout()["solver_fallbacks"] == 0     # the cyclic-reduction kernels did the work (a failed pivot would send the step to the sequential solver silently)
    g.set_switch("SOLVER_SEQ", 1)
    g.evaluate(x, normal_eq=True, dense=False, residuals=False)
    ds, ms = g.solve_step(RADIUS, True)
    g.close()
    assert np.all(np.isfinite(d)) and np.abs(d).max() > 0
    assert np.abs(d - ds).max() <= 1e-7 * max(1.0, np.abs(ds).max())
    assert abs(m - ms) <= 1e-8 * abs(ms)

