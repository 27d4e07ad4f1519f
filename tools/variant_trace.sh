#!/bin/bash
# tools/variant_trace.sh "PROBE CMD" PATTERN TAG... — on the GPU box: run the probe under rocprofv3 --kernel-trace once per variant library (tools/build_variant.sh; "base" = the
# regular build) and print the kernel-summary lines that match PATTERN.
CMD=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for T in "$@"; do
  LIB=lvi-exc_amd/liblvx_var_$T.so; [ "$T" = "base" ] && LIB=lvi-exc_amd/liblvx.so
  rm -rf gpurun_out/vt_$T
  LVX_LIB=$LIB rocprofv3 --kernel-trace -d gpurun_out/vt_$T -o kt -- $CMD > gpurun_out/vt_$T.log 2>&1
  echo "== $T"; python tools/rocpd_summary.py $(find gpurun_out/vt_$T -name "*.db" | head -1) | grep -E "$PAT" | cut -c1-60,95-170
  rm -rf gpurun_out/vt_$T
done
