python -m pytest tests/test_gpu_solver.py tests/test_gpu_shared.py -m gpu -x -q 2>&1 | tail -2
LVX_SOLVER_TIMING=1 python tools/lm_scale_probe.py 2>&1 | grep "sync\|lm 4" | tail -3
