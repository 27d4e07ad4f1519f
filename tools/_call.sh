cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_shapes.py tests/test_gpu_deterministic.py -m gpu -q -x 2>&1 | tail -3
LVX_SERIAL=1 LVX_LIB=lvi-exc_amd/liblvx_kt_GyroAcc.so python bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | grep "^IKT" | tail -3
for e in "X=1" "X=2"; do
  echo "== $e"; env $e LVX_BENCH_NOPROF=1 python bench.py --no-secondary --no-cpu-baseline --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"
done
