#!/bin/bash
# tools/build_kt.sh FAM — instrumented library lvi-exc_amd/liblvx_kt_FAM.so: lvx_eval.hip with the per-phase cycle counters of k_family_mfma<FAM>
# (LVX_KTIME), the other objects from the regular build.  Use: LVX_LIB=lvi-exc_amd/liblvx_kt_FAM.so python bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline
set -e
FAM=${1:-SurfAcc}
cd "$(dirname "$0")/../lvi-exc_amd"
python build.py > /dev/null
TAG=$(echo "$FAM" | tr -cd 'A-Za-z0-9')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result -ffp-contract=fast -DLVX_KTIME "-DLVX_KTIME_FAM=$FAM" -c csrc/lvx_eval.hip -o /tmp/lvx_eval_kt_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblvx_kt_$TAG.so /tmp/lvx_eval_kt_$TAG.o csrc/lvx_api.o csrc/lvx_bcr.o csrc/lvx_solver.o csrc/lvx_upstream.o -ldl -Wl,-rpath,/opt/rocm/lib
echo liblvx_kt_$TAG.so
