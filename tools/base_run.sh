cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_gpu_tests.log 2>&1; tail -3 gpurun_out/r03c_gpu_tests.log
rm -rf gpurun_out/r03c_lm
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r03c_lm -o kt -- python tools/lm_iter_probe.py 8 > gpurun_out/r03c_lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r03c_lm -name "*.db" | head -1) > gpurun_out/r03c_kernel_stats_lm_iteration.txt
tail -4 gpurun_out/r03c_lm.log
timeout 600 python bench.py > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err; tail -1 gpurun_out/r03c_bench.json | cut -c1-300
