cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in lvi-exc_amd/liblvx.so lvi-exc_amd/liblvx_kt_bcr.so; do
rm -rf gpurun_out/exp_lm
LVX_LIB=$lib timeout 300 rocprofv3 --kernel-trace -d gpurun_out/exp_lm -o kt -- python tools/solve_once.py 3 > gpurun_out/exp_lm.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/exp_lm -name "*.db" | head -1) | grep "trsm_lds\|potrf_reg\|KERNEL"
done
