#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests/test_gpu_solver.py tests/test_gpu_solver_sizes.py tests/test_gpu_shared.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
python tools/probes/stage_times.py 2>&1 | grep trajInit
