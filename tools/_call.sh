cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for e in "X=1" "LVX_SINGLE_BUFFER=1" "X=2" "LVX_SINGLE_BUFFER=1"; do
  echo "== $e"; env $e LVX_BENCH_NOPROF=1 python bench.py --no-secondary --no-cpu-baseline --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), d['config']['cost'])"
done
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python tools/bsum.py gpurun_out/b.json
python tools/stress_parity.py 1500 2>&1 | grep -E "worst|MISMATCH" | tail -3
