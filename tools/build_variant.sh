#!/bin/bash
# tools/build_variant.sh FILE TAG "-DX=1 ..." — experiment library lvi-exc_amd/liblvx_var_TAG.so: csrc/FILE.hip rebuilt with extra defines, the other objects from the
# regular build.  Use: LVX_LIB=lvi-exc_amd/liblvx_var_TAG.so python tools/probes/...   (same sources, a measurement switch compiled in; never shipped: *.so is git-ignored)
set -e
FILE=$1; TAG=$2; DEFS=$3
cd "$(dirname "$0")/../lvi-exc_amd"
python build.py > /dev/null
CONTRACT=fast; [ "$FILE" = "lvx_upstream" ] && CONTRACT=off
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result -ffp-contract=$CONTRACT $DEFS -c csrc/$FILE.hip -o /tmp/${FILE}_var_$TAG.o
OBJS=""
for f in lvx_api lvx_bcr lvx_eval lvx_solver lvx_upstream; do
  if [ "$f" = "$FILE" ]; then OBJS="$OBJS /tmp/${FILE}_var_$TAG.o"; else OBJS="$OBJS csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblvx_var_$TAG.so $OBJS -ldl -Wl,-rpath,/opt/rocm/lib
echo liblvx_var_$TAG.so
