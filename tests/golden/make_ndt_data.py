"""Fixture generator: the two real LiDAR scans the reference ships with ndt_omp (BSD-licensed demo data: /root/reference/src/ndt_omp/data/251370668.pcd and
251371071.pcd, the target and source cloud of apps/align.cpp:25-52) as point ARRAYS — tests/golden/ndt_data_<stamp>.npz, xyzi float32 [n][4], bit for bit.
Run in the build container (the reference tree does not exist on the GPU box):  python tests/golden/make_ndt_data.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/src/ndt_omp/data"


def read_pcd(path):
    """Minimal PCD v0.7 reader: `DATA binary` (or ascii), float32 fields x y z intensity."""
    with open(path, "rb") as f:
        hdr = {}
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if not line or line.startswith("#"):
                continue
            k, _, v = line.partition(" ")
            hdr[k] = v.split()
            if k == "DATA":
                break
        fields, n = hdr["FIELDS"], int(hdr["POINTS"][0])
        assert hdr["SIZE"] == ["4"] * len(fields) and hdr["TYPE"] == ["F"] * len(fields) and hdr["COUNT"] == ["1"] * len(fields), hdr
        if hdr["DATA"][0] == "binary":
            a = np.frombuffer(f.read(4 * len(fields) * n), dtype="<f4").reshape(n, len(fields))
        else:
            a = np.loadtxt(f, dtype=np.float32).reshape(n, len(fields))
    out = np.zeros((n, 4), np.float32)
    for j, name in enumerate(("x", "y", "z", "intensity")):
        if name in fields:
            out[:, j] = a[:, fields.index(name)]
    return out


if __name__ == "__main__":
    for stamp in ("251370668", "251371071"):
        xyzi = read_pcd(os.path.join(SRC, stamp + ".pcd"))
        dst = os.path.join(HERE, "ndt_data_%s.npz" % stamp)
        np.savez_compressed(dst, xyzi=xyzi)
        print(dst, xyzi.shape, "finite:", int(np.isfinite(xyzi[:, :3]).all(axis=1).sum()), "bbox", xyzi[:, :3].min(0), xyzi[:, :3].max(0), os.path.getsize(dst), "bytes", file=sys.stderr)
