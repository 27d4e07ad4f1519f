python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/small_latency.py | grep async
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})"; done
