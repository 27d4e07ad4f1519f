#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_solver.py tests/test_gpu_solver_sizes.py tests/test_gpu_shared.py tests/test_gpu_converge.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
python tools/lm_iter_probe.py 6 2>&1 | grep "^lm"
LVX_DETERMINISTIC=1 python tools/lm_iter_probe.py 6 2>&1 | grep "^lm"
