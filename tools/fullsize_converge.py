"""Probe: config 4 through the reference's stage schedule to an LM termination.  Usage: python tools/fullsize_converge.py [scale] [legacy] [sparse] [verbose]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import stages as cs  # noqa: E402
import synth  # noqa: E402

if __name__ == "__main__":
    scale = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
    P = synth.make_bench_problem(seed=4, n_imu=200_000 // scale, n_surfel=1_000_000 // scale, n_reproj=50_000 // scale, n_planes=max(32, 2000 // scale),
                                 tracks="sparse" if "sparse" in sys.argv else "orb")
    x, log = cs.run_stages_gpu(P, P["state0"], legacy="legacy" in sys.argv, verbose=1 if "verbose" in sys.argv else 0)
    for name, s, dt in log:
        print("%-20s it %3d ok %3d %-20s cost %.6e -> %.6e  %.3f s" % (name, s["iterations"], s["successful_steps"], s["termination"], s["initial_cost"], s["final_cost"], dt))
    print(cs.extrinsic_errors(x, P["state_true"], P["n_knots"]))
