#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_upstream.py tests/test_gpu_realdata.py tests/test_golden.py tests/test_gpu_pipeline_oracle.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/upstream_bench.py 2>&1 | grep "voxel\|data_assoc"
bash tools/_call.sh 2>&1 | grep "k_vx_\|k_surfel_extract"
