cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_converge.py tests/test_gpu_shared.py tests/test_host_estimator.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -2
rocprofv3 --kernel-trace -d gpurun_out/lm -o kt -- python tools/lm_scale_probe.py > gpurun_out/lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/lm -name "*.db" | head -1) > gpurun_out/lm_stats.txt
grep "potrf\|back_level\|trsm_reg<true" gpurun_out/lm_stats.txt | cut -c1-60,90-150
python bench.py --steps 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']; print(s['lm_iteration']['ms_per_iteration'], s['converged_solve']['seconds'])"
