cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python tools/bsum.py gpurun_out/b.json
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o pmc -- python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_w gpurun_out/pmc_w 2>/dev/null | grep -A3 "k_clear\|fold_border_rows" | cut -c1-140
