cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LVX_SERIAL=1 LVX_LIB=lvi-exc_amd/liblvx_kt_GyroAcc.so python bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | grep "^IKT" | tail -3
LVX_IMU_MFMA=1 LVX_SERIAL=1 LVX_LIB=lvi-exc_amd/liblvx_kt_GyroAcc.so python bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | grep "^IKT" | tail -2
