cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python tools/stress_parity.py 3000) 2>&1 | tail -4
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -3; done
