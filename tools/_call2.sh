#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_upstream.py tests/test_gpu_pipeline_oracle.py tests/test_gpu_pipeline.py tests/test_gpu_realdata.py tests/test_golden.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
timeout 600 python tools/upstream_bench.py 2>&1 | grep "data_assoc"
