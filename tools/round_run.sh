# tools/round_run.sh TAG — on the GPU box: the whole evidence set of a state: GPU tests, bench line, kernel traces (concurrent + serial), HBM traffic counters,
# SQ counters, timeline, LM-iteration trace; everything under gpurun_out/TAG_*
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_gpu_tests.log | tail -2
bash tools/profile_round.sh $TAG
bash tools/pmc_sq.sh; cp gpurun_out/pmc_sq_summary.txt gpurun_out/${TAG}_pmc_sq.txt
python tools/make_traffic_json.py gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE "profiles/${TAG}_pmc_hbm_traffic.txt" gpurun_out/pmc_sq2 "profiles/${TAG}_pmc_sq.txt" > gpurun_out/${TAG}_latest_traffic.json
python tools/rocpd_timeline.py $(find gpurun_out/${TAG}_kt -name "*.db" | head -1) 12 > gpurun_out/${TAG}_timeline.txt
rm -rf gpurun_out/${TAG}_lm
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_lm -o kt -- python tools/lm_iter_probe.py 8 > gpurun_out/${TAG}_lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_lm -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_lm_iteration.txt
grep "^lm \|solve_step" gpurun_out/${TAG}_lm.log | cut -c1-120
rm -rf gpurun_out/${TAG}_lm gpurun_out/${TAG}_kt gpurun_out/${TAG}_kts gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
