#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/probes/stage_times.py 2>&1 | grep trajInit
timeout 1500 python -m pytest tests/test_gpu_eval.py tests/test_gpu_shapes.py tests/test_gpu_fullsize_oracle.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -3
