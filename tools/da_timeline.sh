#!/bin/bash
# tools/da_timeline.sh — on the GPU box: kernel timeline of one lvx_data_association round (from a rocprofv3 kernel trace of tools/upstream_bench.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/upkt
rocprofv3 --kernel-trace -d gpurun_out/upkt -o kt -- python tools/upstream_bench.py 5 > gpurun_out/upkt.log 2>&1
DB=$(find gpurun_out/upkt -name "*.db" | head -1)
python - $DB <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_lidar_pose" in r[0]]
a, b = idx[-2], idx[-1]
t0 = rows[a][1]
for r in rows[a:b]:
    print("%-70s %9.1f %8.1f" % (r[0][:70], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
PY
rm -rf gpurun_out/upkt
