"""GPU parity of the upstream point-cloud kernels vs the serial CPU oracle (oracle/orc_upstream.cpp).

scanRegistration: bit-exact (float curvature bits, labels, sorted indices, the four index lists) on tie-free sweeps.
voxel grid: leaf keys / counts / point lists exact; mean, cov 1e-12 rel; eigenvalues 1e-10 rel; eigenvectors up to sign;
inverse covariance 1e-8 rel.  DIRECT7 lookup and surfel association: exact integers.
"""
import numpy as np
import pytest

import lvx
import synth
from oracle import oracle as O
from upstream_checks import check_voxels as _check_voxels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = lvx.Context(0)
    yield c
    c.close()


def _check_scanreg(ctx, pts, n_rings, min_range, strict=True):
    ro = O.scan_register(pts, n_rings, min_range)
    rg = lvx.scan_register(ctx, pts, n_rings, min_range)
    assert rg["n"] == ro["n"]
    assert np.array_equal(rg["scan_start"], ro["scan_start"]) and np.array_equal(rg["scan_end"], ro["scan_end"])
    assert np.array_equal(rg["cloud"].view(np.uint32), ro["cloud"].view(np.uint32))
    assert np.array_equal(rg["curvature"].view(np.uint32), ro["curvature"].view(np.uint32))
    if strict:
        for k in ("label", "picked", "sort_ind", "sharp", "less_sharp", "flat", "less_flat"):
            assert np.array_equal(rg[k], ro[k]), k
    else:   # equal curvatures inside a sector: std::sort (unstable) may order them differently -> compare modulo tie permutation
        c = ro["curvature"]
        assert np.array_equal(c[rg["sort_ind"]].view(np.uint32), c[ro["sort_ind"]].view(np.uint32))
        assert np.array_equal(np.sort(rg["sort_ind"]), np.sort(ro["sort_ind"]))
    return ro


@pytest.mark.parametrize("seed", [1, 3, 4])
def test_scan_register_vlp16_bit_exact(ctx, seed):
    pts = synth.make_vlp16_sweep(seed=seed)
    ro = _check_scanreg(ctx, pts, 16, 0.3)
    c = ro["curvature"]
    for i in range(16):            # the comparison is only meaningful on tie-free sectors (std::sort is unstable)
        s, e = ro["scan_start"][i], ro["scan_end"][i]
        for j in range(6):
            sp, ep = s + (e - s) * j // 6, s + (e - s) * (j + 1) // 6 - 1
            assert len(np.unique(c[sp:ep + 1])) == ep - sp + 1
    assert len(ro["sharp"]) > 100 and len(ro["flat"]) > 300


def _tie_sectors(ro):
    c, n = ro["curvature"], 0
    for i in range(len(ro["scan_start"])):
        s, e = ro["scan_start"][i], ro["scan_end"][i]
        for j in range(6):
            sp, ep = s + (e - s) * j // 6, s + (e - s) * (j + 1) // 6 - 1
            n += len(np.unique(c[sp:ep + 1])) < ep - sp + 1
    return n


@pytest.mark.parametrize("kw,min_tie_sectors", [(dict(seed=2), 1), (dict(seed=1, range_quantum=0.002, noise=0.003), 1), (dict(seed=1, xyz_quantum=0.004, noise=0.005), 20),
                                                (dict(seed=7, xyz_quantum=0.01, noise=0.0), 60), (dict(seed=8, xyz_quantum=0.02, noise=0.001, n_az=900), 60)])
def test_scan_register_with_curvature_ties_is_bit_exact(ctx, kw, min_tie_sectors):
    """Equal curvatures inside a sector (quantised ranges; coordinates on a millimetre / centimetre lattice): the reference's std::sort (scanRegistration.cpp:327) is
    unstable, so which of the tied points is picked first is libstdc++ introsort's business.  The kernel restates it on one lane for such sectors
    (lvx_stdsort.h) — sorted indices, labels, picked flags and the four lists are compared STRICTLY, as on tie-free sweeps."""
    pts = synth.make_vlp16_sweep(**kw)
    ro = _check_scanreg(ctx, pts, 16, 0.3, strict=True)
    assert _tie_sectors(ro) >= min_tie_sectors


def test_scan_register_batch_equals_sweep_by_sweep(ctx):
    """lvx_scan_register_batch: sweeps of different sizes in one call — tie-free, with curvature ties, short rings, an EMPTY sweep, one ring only — every sweep's
    outputs bit-exact against the oracle run on that sweep alone, and the less-flat down-sampling addressable per sweep."""
    one_ring = synth.make_vlp16_sweep(seed=5, n_az=300); one_ring["ring"][:] = 3
    sweeps = [synth.make_vlp16_sweep(seed=1), synth.make_vlp16_sweep(seed=7, xyz_quantum=0.01, noise=0.0), synth.make_vlp16_sweep(seed=4, n_az=40), synth.make_vlp16_sweep(seed=3)[:0],
              one_ring, synth.make_vlp16_sweep(seed=3, n_az=900), synth.make_vlp16_sweep(seed=9, n_az=1200, xyz_quantum=0.004, noise=0.005)]
    res = lvx.scan_register_batch(ctx, sweeps, 16, 0.3)
    assert len(res) == len(sweeps)
    for k, (pts, rg) in enumerate(zip(sweeps, res)):
        ro = O.scan_register(pts, 16, 0.3)
        assert rg["n"] == ro["n"], k
        if len(pts):
            assert np.array_equal(rg["scan_start"], ro["scan_start"]) and np.array_equal(rg["scan_end"], ro["scan_end"]), k
        assert np.array_equal(rg["cloud"].view(np.uint32), ro["cloud"].view(np.uint32)) and np.array_equal(rg["curvature"].view(np.uint32), ro["curvature"].view(np.uint32)), k
        for key in ("label", "picked", "sort_ind", "sharp", "less_sharp", "flat", "less_flat"):
            assert np.array_equal(rg[key], ro[key]), (k, key)
    # device-resident variant: points uploaded once by the caller, results stay in the context, one sweep fetched on demand
    import torch
    allp = np.concatenate(sweeps)
    off = np.concatenate([[0], np.cumsum([len(p) for p in sweeps])]).astype(np.int32)
    pd = torch.from_numpy(allp.view(np.uint8).reshape(-1)).to("cuda")
    nk, cnt = lvx.scan_register_batch_d(ctx, pd.data_ptr(), off, 16, 0.3)
    assert list(nk) == [r_["n"] for r_ in res] and [list(c_) for c_ in cnt] == [[len(r_[k]) for k in ("sharp", "less_sharp", "flat", "less_flat")] for r_ in res]
    for k in (1, 6, 3):
        rg = lvx.scan_register_get(ctx, k, len(sweeps[k]), 16)
        for key in ("cloud", "curvature", "label", "picked", "sort_ind", "sharp", "less_sharp", "flat", "less_flat"):
            assert np.array_equal(rg[key], res[k][key]), (k, key)
    for k in (0, 5):
        ro = O.scan_register(sweeps[k], 16, 0.3)
        lf = ro["less_flat"]
        want = np.concatenate([O.voxelgrid_xyzi(ro["cloud"][lf[(lf >= ro["scan_start"][r] - 5) & (lf <= ro["scan_end"][r] + 5)]], 0.2) for r in range(16)])
        got, rc, n = lvx.scan_less_flat_downsample(ctx, 16, len(sweeps[k]), 0.2, sweep=k)
        assert n == len(want) and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_scan_register_edge_cases(ctx):
    pts = synth.make_vlp16_sweep(seed=4, n_az=40)           # rings too short for 6 sectors of >= 1 point after the +-5 margin
    _check_scanreg(ctx, pts, 16, 0.3)
    pts = synth.make_vlp16_sweep(seed=5, n_az=300)
    pts["ring"][:] = 3                                        # everything on one ring, 15 rings empty
    _check_scanreg(ctx, pts, 16, 0.3)
    r = lvx.scan_register(ctx, pts[:0], 16, 0.3)
    assert r["n"] == 0 and len(r["sharp"]) == 0
    pts = synth.make_vlp16_sweep(seed=6, n_az=200)
    _check_scanreg(ctx, pts, 16, 100.0)                       # min range removes every point


def test_voxel_build_and_lookup_config2(ctx):
    cloud = synth.make_voxel_cloud(seed=2, n=100_000)
    vo = O.voxel_build(cloud, 0.5)
    vg = lvx.voxel_build(ctx, cloud, 0.5)
    _check_voxels(vg, vo)
    q = synth.rigid_move(cloud)
    ids7 = O.voxel_lookup7(vo, q, 0.5)
    assert np.array_equal(lvx.voxel_lookup7(ctx, q), ids7)
    assert np.array_equal(lvx.voxel_lookup1(ctx, q), ids7[:, 0])      # getNeighborhoodAtPoint1 = the zero displacement of DIRECT7


def test_voxel_build_large_cloud_and_long_leaves(ctx):
    """> 500 k points take the 1 024-position tiles of k_vx_leaf (four sorted positions per thread); a coarse leaf size makes leaves of thousands of points that
    cross tiles and leave the staged range (their owner gathers the rest itself); non-finite points in between."""
    cloud = synth.tile_voxel_cloud(synth.make_voxel_cloud(seed=3, n=100_000), 6)
    cloud[::1013, 1] = np.nan
    for leaf in (0.5, 4.0):
        vo = O.voxel_build(cloud, leaf)
        vg = lvx.voxel_build(ctx, cloud, leaf)
        _check_voxels(vg, vo)
    assert vo["leaf_n"].max() > 1500
    big = synth.tile_voxel_cloud(synth.make_voxel_cloud(seed=5, n=100_000), 12)   # 1.2 M points: above the own radix sort's range, rocPRIM's sort feeds the same leaf kernel
    vo, vg = O.voxel_build(big, 0.5), lvx.voxel_build(ctx, big, 0.5)
    _check_voxels(vg, vo)
    small = synth.make_voxel_cloud(seed=9, n=30_000)          # the one-position-per-thread tiles with long leaves
    vo, vg = O.voxel_build(small, 6.0), lvx.voxel_build(ctx, small, 6.0)
    _check_voxels(vg, vo)
    assert vo["leaf_n"].max() > 600
    # extents of 300 x 300 x 250 cells: the cell table grows past 2^24 entries, the keys need four 8-bit digits (an even number of sort passes)
    rng = np.random.default_rng(11)
    wide = np.zeros((20_000, 4), np.float32)
    wide[:, :3] = rng.uniform([0, 0, 0], [300, 300, 250], (20_000, 3))
    wide[:6000, :3] = wide[6000:12000, :3] + rng.normal(0, 0.2, (6000, 3)).astype(np.float32)
    ctx2 = lvx.Context()
    vo, vg = O.voxel_build(wide, 1.0), lvx.voxel_build(ctx2, wide, 1.0)
    _check_voxels(vg, vo)
    assert int(np.prod(vg["grid"][6:9])) > (1 << 24)


def test_voxel_edge_cases(ctx):
    cloud = synth.make_voxel_cloud(seed=7, n=5000)
    cloud[::97, 0] = np.nan
    cloud[5::131, 2] = np.inf
    vo = O.voxel_build(cloud, 1.0)
    vg = lvx.voxel_build(ctx, cloud, 1.0)
    _check_voxels(vg, vo)
    far = cloud.copy(); far[:, :3] += 500.0                   # queries outside the grid
    far[::7, 0] = np.nan
    assert np.array_equal(lvx.voxel_lookup7(ctx, far), O.voxel_lookup7(vo, far, 1.0))
    assert np.array_equal(lvx.voxel_lookup1(ctx, far), O.voxel_lookup7(vo, far, 1.0)[:, 0])
    one = np.array([[0.1, 0.2, 0.3, 0.0]] * 7, np.float32)    # a single voxel, 7 coincident points: singular covariance
    vo1, vg1 = O.voxel_build(one, 0.5), lvx.voxel_build(ctx, one, 0.5)
    assert np.array_equal(vg1["leaf_n"], vo1["leaf_n"]) and vg1["n_leaves"] == 1
    assert lvx.voxel_build(ctx, one[:0], 0.5)["n_leaves"] == 0


@pytest.mark.parametrize("n_planes", [1, 400, 3000])
def test_surfel_assoc_exact(ctx, n_planes):
    scan, p4, bmin, bmax = synth.make_assoc_problem(seed=5, n_planes=n_planes)
    fo = O.surfel_assoc(scan, p4, bmin, bmax, 0.05, 2)
    fg = lvx.surfel_assoc(ctx, scan, p4, bmin, bmax, 0.05, 2)
    assert np.array_equal(fg, fo)
    if n_planes >= 400:
        assert (fo >= 0).sum() > 50


def test_surfel_assoc_batch_and_chronological_emission(ctx):
    """S scans per launch + the SurfelPoint emission of surfel_association.cpp:141-158 (column-major, timestamp == 0 skipped), all on the device."""
    scans, raws, fo, eo = [], [], [], []
    p4 = bmin = bmax = None
    for s in range(3):
        scan, p4s, bmins, bmaxs = synth.make_assoc_problem(seed=40 + s, n_planes=300)
        if p4 is None:
            p4, bmin, bmax = p4s, bmins, bmaxs          # one surfel map, several scans
        H, W = scan.shape[0], scan.shape[1]
        raw = np.zeros((H, W), dtype=lvx.POINT_XYZIT)
        rng = np.random.default_rng(s)
        raw["x"], raw["y"], raw["z"] = scan[..., 0] + 1.0, scan[..., 1] - 2.0, scan[..., 2] + 0.5       # some other frame
        raw["timestamp"] = 100.0 + 0.1 * s + np.tile(np.arange(W) / W * 0.1, (H, 1))
        raw["timestamp"][rng.random((H, W)) < 0.05] = 0.0                                                  # dropped returns
        f = O.surfel_assoc(scan, p4, bmin, bmax, 0.05, 2)
        scans.append(scan); raws.append(raw); fo.append(f); eo.append(O.surfel_emit(f, scan, raw))
    fg, eg = lvx.surfel_assoc_emit(ctx, np.stack(scans), np.stack(raws), p4, bmin, bmax, 0.05, 2)
    assert np.array_equal(fg, np.stack(fo))
    assert list(eg["counts"]) == [len(e["t"]) for e in eo] and sum(eg["counts"]) > 50
    for k in ("pt", "pt_map", "t", "plane"):
        assert np.array_equal(eg[k], np.concatenate([e[k] for e in eo]))
    assert (eg["t"] != 0).all()


def test_surfel_map_prepared_once(ctx):
    """lvx_surfel_map_prepare_d: the association grid of a surfel table built once (setSurfelMap), batches of scans associated against it by pointer; a call
    with another table rebuilds and invalidates it."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    scans, want = [], []
    p4 = bmin = bmax = None
    for s in range(4):
        scan, p4s, bmins, bmaxs = synth.make_assoc_problem(seed=60 + s, n_planes=250)
        if p4 is None:
            p4, bmin, bmax = p4s, bmins, bmaxs
        scans.append(scan); want.append(O.surfel_assoc(scan, p4, bmin, bmax, 0.05, 2))
    H, W, P = scans[0].shape[0], scans[0].shape[1], len(p4)
    sc = torch.from_numpy(np.ascontiguousarray(np.stack(scans), np.float32)).to(dev)
    pl = torch.from_numpy(np.concatenate([p4.ravel(), bmin.ravel(), bmax.ravel()])).to(dev)
    fl = torch.empty((4, H * W), dtype=torch.int32, device=dev)
    l = ctx._l
    ctx._ck(l.lvx_surfel_map_prepare_d(ctx._h, C.c_int(P), C.c_void_p(pl.data_ptr())))
    for _ in range(2):   # twice against the prepared map
        fl.fill_(7)
        ctx._ck(l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(4), C.c_int(H), C.c_int(W), C.c_void_p(sc.data_ptr()), C.c_int(P), C.c_void_p(pl.data_ptr()), C.c_double(0.05), C.c_int(2), C.c_void_p(fl.data_ptr())))
        ctx.synchronize()
        assert np.array_equal(fl.cpu().numpy().reshape(4, H, W), np.stack(want))
    # another table (a copy with fewer planes): rebuilt for it, and the prepared state is gone — the first table is rebuilt on its next use
    P2 = P // 2
    pl2 = torch.from_numpy(np.concatenate([p4[:P2].ravel(), bmin[:P2].ravel(), bmax[:P2].ravel()])).to(dev)
    ctx._ck(l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(4), C.c_int(H), C.c_int(W), C.c_void_p(sc.data_ptr()), C.c_int(P2), C.c_void_p(pl2.data_ptr()), C.c_double(0.05), C.c_int(2), C.c_void_p(fl.data_ptr())))
    ctx.synchronize()
    assert np.array_equal(fl.cpu().numpy().reshape(4, H, W), np.stack([O.surfel_assoc(s_, p4[:P2], bmin[:P2], bmax[:P2], 0.05, 2) for s_ in scans]))
    ctx._ck(l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(4), C.c_int(H), C.c_int(W), C.c_void_p(sc.data_ptr()), C.c_int(P), C.c_void_p(pl.data_ptr()), C.c_double(0.05), C.c_int(2), C.c_void_p(fl.data_ptr())))
    ctx.synchronize()
    assert np.array_equal(fl.cpu().numpy().reshape(4, H, W), np.stack(want))
    # lifetime: prepare -> release -> the SAME address and plane count now hold another table (what a caching allocator hands out): no stale grid
    ctx._ck(l.lvx_surfel_map_prepare_d(ctx._h, C.c_int(P), C.c_void_p(pl.data_ptr())))
    ctx._ck(l.lvx_surfel_map_release(ctx._h))
    _, p4b, bminb, bmaxb = synth.make_assoc_problem(seed=77, n_planes=250)
    assert len(p4b) == P and not np.array_equal(p4b, p4)
    pl.copy_(torch.from_numpy(np.concatenate([p4b.ravel(), bminb.ravel(), bmaxb.ravel()])))
    ctx._ck(l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(4), C.c_int(H), C.c_int(W), C.c_void_p(sc.data_ptr()), C.c_int(P), C.c_void_p(pl.data_ptr()), C.c_double(0.05), C.c_int(2), C.c_void_p(fl.data_ptr())))
    ctx.synchronize()
    assert np.array_equal(fl.cpu().numpy().reshape(4, H, W), np.stack([O.surfel_assoc(s_, p4b, bminb, bmaxb, 0.05, 2) for s_ in scans]))


def test_association_work_buffer_is_private(ctx):
    """The association's hit bitmasks are cleared once per shape and left clean by the selection kernel: nothing else may write that buffer.  De-skew and the
    batched pose evaluation upload the state vector into a context buffer between two associations of the same shape (what every refinement round of
    DataAssociation does) — the second association must be unaffected."""
    P = synth.make_problem(seed=33, duration=1.5, n_surfel=50, n_planes=4, n_landmarks=40, n_camsurf=0)
    scan, p4, bmin, bmax = synth.make_assoc_problem(seed=61, n_planes=250)
    want = O.surfel_assoc(scan, p4, bmin, bmax, 0.05, 2)
    assert np.array_equal(lvx.surfel_assoc(ctx, scan, p4, bmin, bmax, 0.05, 2), want)
    ctx.set_spline(P["t0"], P["dt"], P["n_knots"])
    state = P["state_true"].copy()
    state[:] = np.where(np.arange(len(state)) % 2 == 0, -1.0, state)      # bit patterns with every mask bit set somewhere
    raw = np.zeros(64, lvx.POINT_XYZIT); raw["timestamp"] = P["t0"] + 0.3
    for _ in range(2):
        lvx.eval_lidar_pose(ctx, P["state_true"], np.array([P["t0"] + 0.2]))
        try:
            lvx.undistort(ctx, state, raw, [0, 0, 0, 1.0], [0, 0, 0.0])
        except lvx.LvxError:
            pass
        assert np.array_equal(lvx.surfel_assoc(ctx, scan, p4, bmin, bmax, 0.05, 2), want)


def test_landmark_plane_association(ctx):
    """associateVisualPointsWithPlanes (surfel_association.cpp:161-214) against the oracle: boxes around some landmarks' map-frame positions,
    overlapping boxes (the highest index stays), a far landmark (rho < 0.05) and one whose box misses by the strict inequality."""
    P = synth.make_problem(seed=33, duration=1.5, n_surfel=50, n_planes=4, n_landmarks=40, n_camsurf=0)
    N, L = P["n_knots"], P["n_landmarks"]
    state = P["state_true"].copy()
    state[7 * N + 32 + 3] = 0.04                                           # beyond 20 m: skipped
    sp = synth.Spline(P["t0"], P["dt"], state[:3 * N].reshape(N, 3), state[3 * N:7 * N].reshape(N, 4))
    qC, pC = state[7 * N + 24:7 * N + 28], state[7 * N + 28:7 * N + 31]
    qL, pL = state[7 * N + 16:7 * N + 20], state[7 * N + 20:7 * N + 23]
    q_LtoC = synth.qmul(synth.qconj(qC), qL); t_LinC = synth.qrot(synth.qconj(qC), pL - pC)
    e0 = sp.eval([P["t_map"]])
    qCG0, pCG0 = synth.qmul(e0["quat"][0], qC), synth.qrot(e0["quat"][0], pC) + e0["pos"][0]
    qL0, tL0 = synth.qmul(qCG0, q_LtoC), synth.qrot(qCG0, t_LinC) + pCG0
    rng = np.random.default_rng(1)
    p4, bmin, bmax = [], [], []
    for l in range(0, L, 2):                                               # a surfel through every second landmark
        ek = sp.eval([P["lm_t0"][l]])
        pc = synth._unproject(P["camera"], P["lm_uv"][l]) / state[7 * N + 32 + l]
        pG = synth.qrot(synth.qmul(ek["quat"][0], qC), pc) + synth.qrot(ek["quat"][0], pC) + ek["pos"][0]
        pM = synth.qrot(synth.qconj(qL0), pG - tL0)
        n_ = rng.standard_normal(3); n_ /= np.linalg.norm(n_)
        off = rng.uniform(-0.12, 0.12)                                     # some inside 2 * radius = 0.1, some outside
        p4.append([*n_, -n_ @ pM + off]); bmin.append(pM - 0.3); bmax.append(pM + 0.3)
    p4.append(p4[0]); bmin.append(bmin[0] - 0.1); bmax.append(bmax[0] + 0.1)      # overlaps surfel 0: the later index wins
    p4, bmin, bmax = np.array(p4), np.array(bmin), np.array(bmax)
    o = O.Oracle(); lvx.load_problem(o, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    lvx.load_problem(ctx, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    ro = O.landmark_assoc(o, state, q_LtoC, t_LinC, P["t_map"], p4, bmin, bmax, 0.05)
    rg = lvx.landmark_assoc(ctx, state, q_LtoC, t_LinC, P["t_map"], p4, bmin, bmax, 0.05)
    assert np.array_equal(rg, ro)
    assert (ro >= 0).sum() >= 5 and (ro == -1).sum() >= L // 2 and ro[3] == -1


def test_surfel_assoc_overlapping_planes_serial_rule(ctx):
    scan, p4, bmin, bmax = synth.make_assoc_problem(seed=6, n_planes=50)
    p4 = np.concatenate([p4, p4]); bmin = np.concatenate([bmin, bmin]); bmax = np.concatenate([bmax, bmax])   # duplicates: higher id must win
    fo = O.surfel_assoc(scan, p4, bmin, bmax, 0.05, 2)
    fg = lvx.surfel_assoc(ctx, scan, p4, bmin, bmax, 0.05, 2)
    assert np.array_equal(fg, fo)
    assert (fo[fo >= 0] >= 50).all()


def test_lidar_pose_and_undistort(ctx):
    """Scan de-skew: per-point spline pose (FP64) then float output, vs the oracle; includes NaN points and timestamps outside the spline."""
    P = synth.make_problem(seed=41, duration=1.5, n_surfel=0, n_planes=1, n_landmarks=0)
    o = O.Oracle(); lvx.load_problem(o, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    c2 = lvx.Context(0); lvx.load_problem(c2, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    s = P["state_true"]
    rng = np.random.default_rng(3)
    t = np.concatenate([rng.uniform(P["t_start"], P["t_end"], 500), [P["t0"] - 1.0, P["t0"] + (P["n_knots"] - 3) * P["dt"], P["t0"]]])
    qo, po, oko = O.eval_lidar_pose(o, s, t)
    qg, pg, okg = lvx.eval_lidar_pose(c2, s, t)
    assert np.array_equal(oko, okg) and oko[:500].all() and not oko[500] and not oko[501] and oko[502]
    assert np.abs(qo[oko] - qg[okg]).max() < 1e-13 and np.abs(po[oko] - pg[okg]).max() < 1e-12
    n = 5000
    raw = np.zeros(n, dtype=lvx.POINT_XYZIT)
    raw["x"], raw["y"], raw["z"] = (rng.uniform(-20, 20, (3, n))).astype(np.float32)
    raw["intensity"] = rng.uniform(0, 255, n).astype(np.float32)
    raw["timestamp"] = rng.uniform(P["t_start"], P["t_end"], n)
    raw["x"][::97] = np.nan
    raw["timestamp"][5::131] = P["t0"] - 3.0
    q0, p0, _ = O.eval_lidar_pose(o, s, [P["t_map"]])
    qGt = synth.qconj(q0[0])
    for cp in (True, False):
        uo = O.undistort(o, s, raw, qGt, p0[0], cp)
        ug = lvx.undistort(c2, s, raw, qGt, p0[0], cp)
        assert np.array_equal(np.isnan(uo), np.isnan(ug))
        m = ~np.isnan(uo)
        assert np.abs(uo[m] - ug[m]).max() <= 4e-6      # float32 outputs of FP64 poses: at most an ulp or two at |x| ~ 30 m
        assert np.array_equal(uo[:, 3], ug[:, 3])
    c2.close()


def test_surfel_map_extraction(ctx):
    """setSurfelMap (deterministic plane-fit variant): same leaves in the same (voxel key) order as the serial restatement, planes to
    rounding, boxes and inlier counts exact; then the extracted planes drive the association kernel."""
    cloud = synth.make_voxel_cloud(seed=2, n=100_000)
    vo = O.voxel_build(cloud, 0.5)
    ro = O.surfel_extract(cloud, vo)
    lvx.voxel_build(ctx, cloud, 0.5, fetch=False)
    rg, n = lvx.surfel_extract(ctx, max_planes=vo["n_leaves"])
    assert n == len(ro["leaf"]) and n > 200
    assert np.array_equal(rg["leaf"], ro["leaf"]) and np.array_equal(rg["n_points"], ro["n_points"])
    assert np.array_equal(rg["n_inliers"], ro["n_inliers"]) and np.array_equal(rg["plane_type"], ro["plane_type"])
    assert np.abs(rg["p4"] - ro["p4"]).max() <= 1e-9 and np.abs(rg["Pi"] - ro["Pi"]).max() <= 1e-9 * np.abs(ro["Pi"]).max()
    assert np.array_equal(rg["box_min"], ro["box_min"]) and np.array_equal(rg["box_max"], ro["box_max"])
    # properties of the reference's definition (surfel_association.cpp:70-80): Pi = -d n, unit normals, at least 20 inliers, at least 10 points
    assert np.abs(np.linalg.norm(rg["p4"][:, :3], axis=1) - 1).max() <= 1e-12
    assert np.abs(rg["Pi"] + rg["p4"][:, 3:4] * rg["p4"][:, :3]).max() <= 1e-12
    assert rg["n_inliers"].min() >= 20 and rg["n_points"].min() >= 10
    # capacity smaller than the number of planes: count still reported, first planes returned
    r2, n2 = lvx.surfel_extract(ctx, max_planes=5)
    assert n2 == n and len(r2) == 5 and np.array_equal(r2["leaf"], rg["leaf"][:5])
    # stricter planarity keeps a subset
    r3, n3 = lvx.surfel_extract(ctx, max_planes=vo["n_leaves"], p_lambda=0.95)
    assert 0 < n3 < n and set(r3["leaf"]) <= set(rg["leaf"])
    # feed the association kernel with the extracted planes (one sweep standing at the origin of the same scene)
    scan = cloud[:16 * 1800].reshape(16, 1800, 4)
    fg = lvx.surfel_assoc(ctx, scan, rg["p4"], rg["box_min"], rg["box_max"], 0.05, 2)
    fo = O.surfel_assoc(scan, ro["p4"], ro["box_min"], ro["box_max"], 0.05, 2)
    assert np.array_equal(fg, fo) and (fg >= 0).sum() > 0


def _ndt_transform(cloud, p6):
    """pcl::transformPointCloud with T = Translation(p0..2) * Rx(p3) * Ry(p4) * Rz(p5) in float (ndt_omp_impl.hpp:139-146)."""
    cx, sx, cy, sy, cz, sz = np.cos(p6[3]), np.sin(p6[3]), np.cos(p6[4]), np.sin(p6[4]), np.cos(p6[5]), np.sin(p6[5])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    R = (Rx @ Ry @ Rz).astype(np.float32)
    out = cloud.copy()
    out[:, :3] = (cloud[:, :3] @ R.T + np.asarray(p6[:3], np.float32)).astype(np.float32)
    return out


def test_ndt_derivatives(ctx):
    """computeDerivatives (DIRECT7): float per-point arithmetic, double accumulation; GPU vs the serial restatement, and the gradient against
    finite differences of the score (loose: the objective is evaluated in float)."""
    cloud = synth.make_voxel_cloud(seed=2, n=60_000)
    src = synth.make_voxel_cloud(seed=2, n=60_000)[::7].copy()           # a sub-sampled scan of the same scene
    vo = O.voxel_build(cloud, 1.0)
    lvx.voxel_build(ctx, cloud, 1.0, fetch=False)
    for p6 in (np.zeros(6), np.array([0.12, -0.07, 0.03, 0.01, -0.015, 0.02])):
        tr = _ndt_transform(src, p6)
        so, go, Ho = O.ndt_derivatives(vo, 1.0, src, tr, p6)
        sg, gg, Hg = lvx.ndt_derivatives(ctx, src, tr, p6)
        assert so > 0 and abs(sg - so) <= 1e-6 * abs(so)        # gauss_d1 < 0: the score is a likelihood to maximise
        assert np.abs(gg - go).max() <= 1e-5 * np.abs(go).max() and np.abs(Hg - Ho).max() <= 1e-5 * np.abs(Ho).max()
        s2, g2, _ = lvx.ndt_derivatives(ctx, src, tr, p6, compute_hessian=False)
        assert s2 == pytest.approx(sg, rel=1e-12) and np.allclose(g2, gg, rtol=1e-12)
    # finite differences of the score along each parameter (translation 1e-3 m, rotation 1e-4 rad)
    p0 = np.array([0.12, -0.07, 0.03, 0.01, -0.015, 0.02])
    _, g0, H0 = lvx.ndt_derivatives(ctx, src, _ndt_transform(src, p0), p0)
    for k in range(6):
        h = 1e-3 if k < 3 else 1e-4
        pp, pm = p0.copy(), p0.copy(); pp[k] += h; pm[k] -= h
        fd = (lvx.ndt_derivatives(ctx, src, _ndt_transform(src, pp), pp, compute_hessian=False)[0] - lvx.ndt_derivatives(ctx, src, _ndt_transform(src, pm), pm, compute_hessian=False)[0]) / (2 * h)
        # the objective jumps whenever a point changes its voxel neighbourhood; rotations move the far points by millimetres per step
        assert abs(fd - g0[k]) <= (0.05 if k < 3 else 0.25) * np.abs(g0).max()
    assert np.abs(H0 - H0.T).max() <= 1e-3 * np.abs(H0).max()
    # empty input
    s0, g00, _ = lvx.ndt_derivatives(ctx, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), p0)
    assert s0 == 0 and not g00.any()


def test_less_flat_voxelgrid_downsample(ctx):
    """The published less-flat cloud (scanRegistration.cpp:425-447): pcl::VoxelGrid 0.2 m per ring, rings concatenated; bit-exact against the
    serial restatement (both average a voxel's points in input order)."""
    pts = synth.make_vlp16_sweep(seed=1)
    r = lvx.scan_register(ctx, pts, 16, 0.3)
    ds, ring_counts, n = lvx.scan_less_flat_downsample(ctx, 16, max_out=len(r["less_flat"]))
    cloud = r["cloud"]
    exp = []
    for ring in range(16):
        idx = r["less_flat"][(r["less_flat"] >= r["scan_start"][ring] - 5) & (r["less_flat"] <= r["scan_end"][ring] + 5)]
        e = O.voxelgrid_xyzi(cloud[idx], 0.2)
        assert ring_counts[ring] == len(e)
        exp.append(e)
    exp = np.concatenate(exp)
    assert n == len(exp) and 0 < n < len(r["less_flat"])
    assert np.array_equal(ds.view(np.uint32), exp.view(np.uint32))
    # a smaller output buffer: the total is still reported
    d2, _, n2 = lvx.scan_less_flat_downsample(ctx, 16, max_out=100)
    assert n2 == n and np.array_equal(d2, ds[:100])
    # coarser leaf => fewer points
    _, _, n3 = lvx.scan_less_flat_downsample(ctx, 16, max_out=10, leaf=1.0)
    assert n3 < n


def test_data_association_map_time_outside_the_trajectory():
    """lvx_data_association keeps the map-time pose on the device (no host hop between k_lidar_pose and the de-skew): an invalid pose makes NaN points, and the call reports
    LVX_E_RANGE at its next synchronisation — with the reference's wording — instead of a surfel map built from garbage; the context is usable afterwards."""
    import ctypes as C
    S = synth.make_sequence(seed=50)
    g = lvx.Context(0)
    g.set_spline(S["t0"], S["dt"], S["n_knots"])
    raw = np.zeros(S["scans"].shape, dtype=lvx.POINT_XYZIT)
    for k in ("x", "y", "z", "timestamp"):
        raw[k] = S["scans"][k]
    g._ck(g._l.lvx_set_scans(g._h, C.c_int(len(raw)), C.c_int(S["H"]), C.c_int(S["W"]), raw.ctypes.data_as(C.c_void_p)))
    st = np.ascontiguousarray(S["state0"], np.float64)
    npl, npt = C.c_int32(0), C.c_int32(0)
    call = lambda t: g._ck(g._l.lvx_data_association(g._h, st.ctypes.data_as(C.c_void_p), C.c_double(t), None, C.byref(npl), C.byref(npt)))
    with pytest.raises(lvx.LvxError) as e:
        call(S["t0"] - 100.0)
    assert e.value.code == lvx.E_RANGE and "map time outside the trajectory" in str(e.value)
    call(S["t_map"])
    assert npl.value > 100 and npt.value > 1000
    first = (npl.value, npt.value)
    call(S["t_map"])
    assert (npl.value, npt.value) == first
    g.close()


def _da_round(g, S, state, opt=None):
    npl, npt = lvx.data_association(g, state, S["t_map"], opt)
    return dict(n=(npl, npt), planes=lvx.get_surfel_map(g, npl), points=lvx.get_surfel_points(g, npt), cloud=lvx.get_scans_in_map(g, len(S["scans"]), S["H"], S["W"]))


def _da_same(a, b):
    assert a["n"] == b["n"]
    assert np.array_equal(a["cloud"].view(np.uint32), b["cloud"].view(np.uint32))
    assert a["planes"].tobytes() == b["planes"].tobytes()
    for k in ("pt", "pt_map", "t", "plane"):
        assert np.array_equal(a["points"][k], b["points"][k]), k


def _da_context(S, sync=False):
    g = lvx.Context(0)
    if sync:
        g.set_switch("DA_SYNC", 1)
    g.set_spline(S["t0"], S["dt"], S["n_knots"])
    raw = np.zeros(S["scans"].shape, dtype=lvx.POINT_XYZIT)
    for k in ("x", "y", "z", "timestamp"):
        raw[k] = S["scans"][k]
    lvx.set_scans(g, raw, S["H"], S["W"])
    return g


def test_data_association_one_stop_rounds_equal_the_four_stop_chain():
    """Round 2 onwards lvx_data_association launches the whole chain over the capacities of the round before and stops at the host once.  Same surfel map,
    same SurfelPoints, same map cloud — bit for bit — as a context that always runs the four-stop chain (LVX_DA_SYNC), at the same state and at a state moved between
    the rounds (the refinement's situation: leaf, plane and list counts all change a little)."""
    S = synth.make_sequence(seed=50)
    rng = np.random.default_rng(5)
    N = S["n_knots"]
    states = [np.ascontiguousarray(S["state0"], np.float64)]
    for _ in range(3):   # the trajectory moves by millimetres / tenths of a milliradian per refinement round
        x = states[-1].copy()
        x[:3 * N] += 2e-3 * rng.standard_normal(3 * N)                      # state = r3_cp[n][3] | so3_cp[n][4] | ... (include/lvx.h)
        q = x[3 * N:7 * N].reshape(N, 4) + 2e-4 * rng.standard_normal((N, 4))
        x[3 * N:7 * N] = (q / np.linalg.norm(q, axis=1, keepdims=True)).ravel()
        states.append(x)
    g, ref = _da_context(S), _da_context(S, sync=True)
    counts = []
    for i, x in enumerate(states + [states[0], states[0]]):
        a, b = _da_round(g, S, x), _da_round(ref, S, x)
        _da_same(a, b)
        counts.append(a["n"])
    runs, repeats = lvx.data_association_stats(g)
    print("rounds:", counts, "one-stop rounds %d, repeated %d" % (runs, repeats))
    assert len(set(counts[:4])) > 1                        # the counts really moved between the rounds
    assert runs == len(counts) - 1 and repeats == 0        # every round after the first took the one-stop chain and none outgrew its capacities
    assert lvx.data_association_stats(ref) == (0, 0)
    assert counts[0][0] > 100 and counts[0][1] > 1000
    g.close(); ref.close()


def test_data_association_outgrown_capacity_repeats_the_round():
    """Capacities learned from a round with few leaves / planes / list entries (a coarse, strict surfel map), then a round with the defaults: every count outgrows what was
    planned, the one-stop attempt is discarded and the four-stop chain gives exactly what a fresh context gives; the next round is one-stop again."""
    S = synth.make_sequence(seed=50)
    x = np.ascontiguousarray(S["state0"], np.float64)
    g, ref = _da_context(S), _da_context(S, sync=True)
    strict = lvx.assoc_default_options(g, ndt_resolution=4.0, min_inliers=60, min_leaf_points=40)
    a0 = _da_round(g, S, x, strict)
    _da_same(a0, _da_round(ref, S, x, strict))
    a1, b1 = _da_round(g, S, x), _da_round(ref, S, x)
    assert a1["n"][0] > 4 * a0["n"][0]
    _da_same(a1, b1)
    assert lvx.data_association_stats(g) == (1, 1)
    _da_same(_da_round(g, S, x), b1)
    assert lvx.data_association_stats(g) == (2, 1)
    # and back to the small map: the capacities shrink when a count falls under a quarter of them — still the same results
    _da_same(_da_round(g, S, x, strict), a0)
    _da_same(_da_round(g, S, x, strict), a0)
    # a map time outside the trajectory on the one-stop chain: the same error, the context stays usable
    with pytest.raises(lvx.LvxError) as e:
        lvx.data_association(g, x, S["t0"] - 100.0)
    assert e.value.code == lvx.E_RANGE and "map time outside the trajectory" in str(e.value)
    _da_same(_da_round(g, S, x), b1)
    g.close(); ref.close()


def test_surfel_assoc_wide_scans_and_double_precision_boxes():
    """W = 4096 columns (128 mask words per ring: two words per bit of the ring's occupancy word, 32 column blocks of the single-launch emission), H = 32 rings, and
    surfel boxes whose bounds are NOT floats (the hit kernel compares the float point with the bounds rounded outwards to floats, which must decide exactly what the
    reference's double comparison decides — some bounds are set to lie between a scan coordinate and its float neighbour)."""
    S, H, W = 4, 32, 4096
    ctx = lvx.Context(0)   # (its own context: lvx_synchronize also reports what an earlier asynchronous evaluation of a shared context left in the device error word)
    rng = np.random.default_rng(3)
    scans, raws = [], []
    p4 = bmin = bmax = None
    for s in range(S):
        scan, p4s, bmins, bmaxs = synth.make_assoc_problem(seed=90 + s, H=H, W=W, n_planes=500)
        if p4 is None:
            p4, bmin, bmax = p4s, bmins.copy(), bmaxs.copy()
        raw = np.zeros((H, W), dtype=lvx.POINT_XYZIT)
        raw["x"], raw["y"], raw["z"] = scan[..., 0], scan[..., 1], scan[..., 2]
        raw["timestamp"] = 50.0 + 0.1 * s + np.tile(np.arange(W) / W * 0.1, (H, 1))
        raw["timestamp"][rng.random((H, W)) < 0.03] = 0.0
        scans.append(scan); raws.append(raw)
    # bounds off the float lattice: a relative 1e-12 nudge either way, and for a tenth of the boxes a bound half a float ulp beside a coordinate of a scan point
    bmin *= 1.0 + 1e-12 * rng.standard_normal(bmin.shape); bmax *= 1.0 + 1e-12 * rng.standard_normal(bmax.shape)
    pts = scans[0].reshape(-1, 4)[:, :3]
    pts = pts[~np.isnan(pts[:, 0])].astype(np.float64)
    for k in range(0, len(p4), 10):
        q = pts[rng.integers(len(pts))]
        a = int(rng.integers(3))
        half_ulp = 0.5 * abs(float(np.spacing(np.float32(q[a]))))
        if k % 20 == 0:
            bmin[k] = q - 0.4; bmax[k] = q + 0.4; bmin[k, a] = q[a] + (half_ulp if k % 40 == 0 else -half_ulp)    # just above / just below the point's coordinate
        else:
            bmin[k] = q - 0.4; bmax[k] = q + 0.4; bmax[k, a] = q[a] + (half_ulp if k % 30 == 0 else -half_ulp)
    assert (bmin.astype(np.float32).astype(np.float64) != bmin).any()
    fo = [O.surfel_assoc(sc, p4, bmin, bmax, 0.05, 2) for sc in scans]
    eo = [O.surfel_emit(f, sc, rw) for f, sc, rw in zip(fo, scans, raws)]
    fg, eg = lvx.surfel_assoc_emit(ctx, np.stack(scans), np.stack(raws), p4, bmin, bmax, 0.05, 2)
    assert np.array_equal(fg, np.stack(fo))
    assert sum((f >= 0).sum() for f in fo) > 500
    assert list(eg["counts"]) == [len(e["t"]) for e in eo]
    for k in ("pt", "pt_map", "t", "plane"):
        assert np.array_equal(eg[k], np.concatenate([e[k] for e in eo]))
    ctx.close()
