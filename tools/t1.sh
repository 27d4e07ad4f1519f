for r in 20 25 28 30 33 40; do
env LVX_SERIAL=1 LVX_CHUNK_R_REP=$r python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('REP=$r', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
