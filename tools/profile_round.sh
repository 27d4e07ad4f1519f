#!/bin/bash
# tools/profile_round.sh TAG — on the GPU box: bench line + rocprofv3 kernel trace + HBM-traffic PMC passes, all under gpurun_out/TAG_*
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_kt -o kt -- python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_kt -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats.txt
LVX_SERIAL=1 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_kts -o kt -- python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_kts.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_kts -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_serial.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_$c -o pmc -- python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE > gpurun_out/${TAG}_pmc_hbm.txt
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-600
