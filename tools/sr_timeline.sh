cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/sr_tl
python tools/probes/sr_probe.py 1 2>&1 | tail -1
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/sr_tl -o kt -- python tools/probes/sr_probe.py 1 > gpurun_out/sr_tl.log 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/sr_tl/**/*.db", recursive=True)[0])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
marks = [i for i, r in enumerate(rows) if "k_sr_gather" in r[0]]
a, b = marks[-2], marks[-1]
t0 = rows[a][1]
for r in rows[a:b]:
    print("%-60s %9.1f %8.1f %9.1f" % (r[0][:60], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3))
print("launches", b - a, "span to next call %.1f us" % ((rows[b][1] - t0) / 1e3))
PY
rm -rf gpurun_out/sr_tl
