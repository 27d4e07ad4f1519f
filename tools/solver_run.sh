# tools/solver_run.sh TAG [pytest -k expr] — on the GPU box: solver tests, then the kernel trace of 8 LM iterations of config 4
TAG=${1:-rXX}; KEXPR=${2:-"solver or fullsize or converge or variants or shared"}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > gpurun_out/${TAG}_solver_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/${TAG}_solver_tests.log | tail -5
rm -rf gpurun_out/${TAG}_lm
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_lm -o kt -- python tools/lm_iter_probe.py 8 > gpurun_out/${TAG}_lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_lm -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_lm_iteration.txt
rm -rf gpurun_out/${TAG}_lm
grep "^lm \|solve_step" gpurun_out/${TAG}_lm.log | cut -c1-120
head -12 gpurun_out/${TAG}_kernel_stats_lm_iteration.txt | cut -c1-60,95-160
