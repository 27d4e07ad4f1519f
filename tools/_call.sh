cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_upstream.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -5
rocprofv3 --kernel-trace -d gpurun_out/up -o kt -- python tools/upstream_bench.py 20 > gpurun_out/up.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/up -name "*.db" | head -1) > gpurun_out/up_stats.txt
grep -v "^W2026\|^E2026" gpurun_out/up.log | tail -2; grep "k_sr_" gpurun_out/up_stats.txt | cut -c1-50,90-150
