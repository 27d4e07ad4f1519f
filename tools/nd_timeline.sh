cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/nd_lm
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/nd_lm -o kt -- python tools/lm_iter_probe.py 6 > gpurun_out/nd_lm.log 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/nd_lm -name "*.db" | head -1) 3 > gpurun_out/nd_timeline.txt
cat gpurun_out/nd_timeline.txt | cut -c1-110
rm -rf gpurun_out/nd_lm
