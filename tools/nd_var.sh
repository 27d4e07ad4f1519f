cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in ${@:-0 1 2 4 8 15}; do
  rm -rf gpurun_out/nd_lm
  LVX_ND_VAR=$V timeout 300 rocprofv3 --kernel-trace -d gpurun_out/nd_lm -o kt -- python tools/lm_iter_probe.py 2 > gpurun_out/nd_lm.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/nd_lm -name "*.db" | head -1) > gpurun_out/nd_var.txt
  echo "== VAR $V"; grep "k_nd_solve\|k_nd_factor" gpurun_out/nd_var.txt | cut -c1-40,95-140
done
rm -rf gpurun_out/nd_lm
