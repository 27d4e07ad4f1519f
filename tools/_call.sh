cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for e in "X=1" "LVX_REF_SIDE=2" "X=2" "LVX_REF_SIDE=2"; do
  echo "== $e"; env $e LVX_BENCH_NOPROF=1 python bench.py --no-secondary --no-cpu-baseline --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"
done
