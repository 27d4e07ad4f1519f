python -m pytest tests/test_gpu_eval.py tests/test_gpu_shapes.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
for v in "LVX_SERIAL=1" ""; do
env $v python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
