#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_shared.py tests/test_gpu_solver.py tests/test_gpu_converge.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -3
