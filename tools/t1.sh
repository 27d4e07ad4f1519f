for gl in 16 32 64; do
sed -i "s/enum { NK = 24, NG = 12, NR = 1, HUB = 0, KPK = 6, LVO = 0, WS = 2, GL = [0-9]*,/enum { NK = 24, NG = 12, NR = 1, HUB = 0, KPK = 6, LVO = 0, WS = 2, GL = $gl,/" lvi-exc_amd/csrc/lvx_eval.hip
python lvi-exc_amd/build.py 2>&1 | grep -E " error" | head -3
LVX_SERIAL=1 python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GL=$gl', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
