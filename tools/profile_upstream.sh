#!/bin/bash
# tools/profile_upstream.sh TAG — on the GPU box: the upstream kernels (scan registration single / batched, voxel build at 100 k / 410 k / 4 M points, DIRECT7 lookup,
# surfel association, NDT derivatives, de-skew, the device-resident DataAssociation round) under rocprofv3: wall times, per-kernel trace, HBM FETCH / WRITE per dispatch.
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/upstream_bench.py 20 > gpurun_out/${TAG}_upstream_bench.txt 2>&1
rm -rf gpurun_out/${TAG}_upkt
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_upkt -o kt -- python tools/upstream_bench.py 10 > gpurun_out/${TAG}_upkt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_upkt -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_upstream.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_uppmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_uppmc_$c -o pmc -- python tools/upstream_bench.py 3 > gpurun_out/${TAG}_uppmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${TAG}_uppmc_FETCH_SIZE gpurun_out/${TAG}_uppmc_WRITE_SIZE > gpurun_out/${TAG}_pmc_hbm_traffic_upstream.txt
rm -rf gpurun_out/${TAG}_upkt gpurun_out/${TAG}_uppmc_FETCH_SIZE gpurun_out/${TAG}_uppmc_WRITE_SIZE
cat gpurun_out/${TAG}_upstream_bench.txt
