#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', round(d['ms_per_step'],4), d['value'], d['kernel_ms'])"; done
timeout 1500 python -m pytest tests/test_gpu_eval.py tests/test_gpu_shapes.py tests/test_gpu_variants.py tests/test_gpu_deterministic.py tests/test_gpu_fullsize_oracle.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -3
