# tools/pmc_lm.sh TAG — HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) of the LM-iteration probe: the solver's kernels
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_lmpmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_lmpmc_$c -o pmc -- python tools/lm_iter_probe.py 3 > gpurun_out/${TAG}_lmpmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${TAG}_lmpmc_FETCH_SIZE gpurun_out/${TAG}_lmpmc_WRITE_SIZE > gpurun_out/${TAG}_pmc_hbm_traffic_lm_iteration.txt
rm -rf gpurun_out/${TAG}_lmpmc_FETCH_SIZE gpurun_out/${TAG}_lmpmc_WRITE_SIZE
grep -A3 "potrf_reg\|trsm_lds\|bcr_schur\|back_level" gpurun_out/${TAG}_pmc_hbm_traffic_lm_iteration.txt | cut -c1-120 | head -30
