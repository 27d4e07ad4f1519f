cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/lmprof
rocprofv3 --kernel-trace -d gpurun_out/lmprof -o lm -- python tools/lm_scale_probe.py > gpurun_out/lmprof.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/lmprof -name "*.db" | head -1) > gpurun_out/lmprof_stats.txt
