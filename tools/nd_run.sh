cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/nd_lm
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/nd_lm -o kt -- python tools/lm_iter_probe.py 8 > gpurun_out/nd_lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/nd_lm -name "*.db" | head -1) > gpurun_out/nd_kernel_stats_lm_iteration.txt
grep "^lm \|solve_step" gpurun_out/nd_lm.log | cut -c1-160
head -30 gpurun_out/nd_kernel_stats_lm_iteration.txt | cut -c1-150
rm -rf gpurun_out/nd_lm
