bash tools/profile_round.sh r01j
bash tools/pmc_sq.sh
cp gpurun_out/pmc_sq_summary.txt gpurun_out/r01j_pmc_sq.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $(find gpurun_out/r01j_kt -name "*.db" | head -1) 12 > gpurun_out/r01j_timeline.txt
rm -rf gpurun_out/r01j_lm
rocprofv3 --kernel-trace -d gpurun_out/r01j_lm -o kt -- python tools/lm_scale_probe.py > gpurun_out/r01j_lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r01j_lm -name "*.db" | head -1) > gpurun_out/r01j_kernel_stats_lm_iteration.txt
tail -3 gpurun_out/r01j_lm.log
