cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for M in ${@:-128 192 256}; do
  rm -rf gpurun_out/nd_lm
  LVX_ND_LEAF=$M timeout 300 rocprofv3 --kernel-trace -d gpurun_out/nd_lm -o kt -- python tools/lm_iter_probe.py 8 > gpurun_out/nd_lm.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/nd_lm -name "*.db" | head -1) > gpurun_out/nd_kernel_stats_lm_iteration_$M.txt
  echo "== LEAF $M"; grep "^lm \|solve_step" gpurun_out/nd_lm.log | cut -c1-100
  grep "k_nd_\|potrf_reg<12>\|k_trsm_lds<12\|potrf_reg<4>\|k_trsm_lds<4\|back_level\|k_bcr_schur<4" gpurun_out/nd_kernel_stats_lm_iteration_$M.txt | cut -c1-60,95-140
done
rm -rf gpurun_out/nd_lm
