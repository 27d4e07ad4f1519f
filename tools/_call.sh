timeout 1200 bash tools/profile_round.sh r02c
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/r02c_lm -o kt -- python tools/lm_scale_probe.py > gpurun_out/r02c_lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r02c_lm -name "*.db" | head -1) > gpurun_out/r02c_kernel_stats_lm_iteration.txt
python tools/rocpd_timeline.py $(find gpurun_out/r02c_kt -name "*.db" | head -1) 12 > gpurun_out/r02c_timeline.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > gpurun_out/r02c_gputest.log; cat gpurun_out/r02c_gputest.log
python -c "import __graft_entry__ as g; g.smoke()"
