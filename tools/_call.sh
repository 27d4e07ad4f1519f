cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/lm -o kt -- python tools/lm_scale_probe.py > gpurun_out/lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/lm -name "*.db" | head -1) > gpurun_out/lm_stats.txt
head -34 gpurun_out/lm_stats.txt | cut -c1-60,90-150
