"""bench.py's Emitter: the JSON line is printed exactly once, and the watchdog prints it (and ends the process with status 0) when a side
measurement never returns — a collective that hangs at N > 1 must not cost the headline line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=60)


def test_emit_once():
    r = _run("import bench; e = bench.Emitter(0); e.out = {'value': 1.0}; e.arm(30); e.disarm(); e.emit(); e.emit()")
    assert r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"value": 1.0}


def test_watchdog_emits_and_exits_when_a_side_measurement_hangs():
    r = _run("import bench, time; e = bench.Emitter(0); e.out = {'value': 2.0}; e.arm(1); time.sleep(30); print('not reached')")
    assert r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 2.0 and "watchdog" in d["secondary"]


def test_other_ranks_exit_silently():
    r = _run("import bench, time; e = bench.Emitter(1); e.out = {'value': 3.0}; e.timer = None\nimport threading\ne.arm(-9); time.sleep(30)")
    assert r.returncode == 0 and r.stdout.strip() == ""
