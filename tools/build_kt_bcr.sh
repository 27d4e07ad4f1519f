#!/bin/bash
# tools/build_kt_bcr.sh [-DFLAG ...] — instrumented library lvi-exc_amd/liblvx_kt_bcr.so: lvx_bcr.hip with per-phase cycle counters (default -DLVX_POTRF_KT),
# the other objects from the regular build.  Use: LVX_LIB=lvi-exc_amd/liblvx_kt_bcr.so python tools/solve_once.py
set -e
FLAGS=${@:--DLVX_POTRF_KT}
cd "$(dirname "$0")/../lvi-exc_amd"
python build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result -ffp-contract=fast $FLAGS -c csrc/lvx_bcr.hip -o /tmp/lvx_bcr_kt.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblvx_kt_bcr.so csrc/lvx_eval.o /tmp/lvx_bcr_kt.o csrc/lvx_solver.o csrc/lvx_upstream.o -L/opt/rocm/lib -lrocblas -lrocsolver -ldl -Wl,-rpath,/opt/rocm/lib
echo liblvx_kt_bcr.so
