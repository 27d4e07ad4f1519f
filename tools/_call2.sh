#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests/test_gpu_solver.py tests/test_gpu_solver_sizes.py tests/test_gpu_shared.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_converge.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
