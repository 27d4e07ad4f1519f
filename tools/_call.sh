cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py -m gpu -q -x 2>&1 | tail -2
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python tools/bsum.py gpurun_out/b.json
for e in "X=1" "X=2"; do
  echo "== $e"; env $e LVX_BENCH_NOPROF=1 python bench.py --no-secondary --no-cpu-baseline --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"
done
