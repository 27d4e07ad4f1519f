#!/bin/bash
# tools/kt_serial.sh TAG [ENV=VAL ...] — on the GPU box: solo kernel durations (LVX_SERIAL=1) of a short bench run through rocprofv3 --kernel-trace
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
env LVX_SERIAL=1 "$@" rocprofv3 --kernel-trace -d gpurun_out/${TAG}_kts -o kt -- python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_kts.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_kts -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_serial.txt
rm -rf gpurun_out/${TAG}_kts
head -14 gpurun_out/${TAG}_kernel_stats_serial.txt | cut -c1-60,90-170
tail -c 400 gpurun_out/${TAG}_kts.log | head -c 300
