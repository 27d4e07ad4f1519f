cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/stress_parity.py 3000 2>&1 | grep -E "worst|MISMATCH" | tail -3
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_converge.py tests/test_gpu_shared.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -2
rocprofv3 --kernel-trace -d gpurun_out/lm -o kt -- python tools/lm_scale_probe.py > gpurun_out/lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/lm -name "*.db" | head -1) > gpurun_out/lm_stats.txt
head -6 gpurun_out/lm_stats.txt | cut -c1-60,90-150
