"""CPU: the oracle's results on the reference's two real scans (tests/golden/ndt_data_*.npz) are pinned by committed checksums of their INTEGER outputs
(leaf keys, point lists, surfel leaves, inlier counts, DIRECT7 ids, the 0.1 m VoxelGrid cloud) — a change of the oracle's upstream restatements that alters
any of them on real data fails here, without a GPU.  The numbers were produced by this oracle (g++ 11.4, -O2 -ffp-contract=off) at the commit that added the
fixtures; tests/test_gpu_realdata.py holds the HIP kernels against the same oracle."""
import os
import zlib

import numpy as np

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())


def test_oracle_on_the_reference_scans():
    tgt = np.load(os.path.join(GOLD, "ndt_data_251370668.npz"))["xyzi"]
    src = np.load(os.path.join(GOLD, "ndt_data_251371071.npz"))["xyzi"]
    want = {"tgt": (2683, 1520, 248, 2768407383, 676534588, 2220440125, 2617669769), "src": (2654, 1453, 242, 3154047074, 338384474, 1982366537, 2151966529)}
    for name, c in (("tgt", tgt), ("src", src)):
        v = O.voxel_build(c, 0.5)                       # lvi.yaml:26
        s = O.surfel_extract(c, v)
        got = (v["n_leaves"], int((v["leaf_n"] >= 6).sum()), len(s["p4"]), crc(v["leaf_key"]), crc(v["point_ids"]), crc(s["leaf"]), crc(s["n_inliers"]))
        assert got == want[name]
    td, sd = O.voxelgrid_xyzi(tgt, 0.1), O.voxelgrid_xyzi(src, 0.1)     # align.cpp:60-69
    v = O.voxel_build(td, 1.0)                                            # align.cpp:85
    assert (len(td), len(sd), v["n_leaves"], crc(td), crc(O.voxel_lookup7(v, sd, 1.0))) == (15772, 15950, 1098, 1973183148, 3020132118)
    score = O.ndt_derivatives(v, 1.0, sd, sd, np.zeros(6))[0]
    assert abs(score - 13682.19599094454) <= 1e-9 * 13682.2
