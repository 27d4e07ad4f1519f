cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "" "LVX_REF_SIDE=0" "LVX_REF_SIDE=2" "LVX_JAC_LATE=1" "LVX_FOLD_ONE=1" "LVX_FOLD_INLINE=1" "LVX_REF_SIDE=0 LVX_FOLD_INLINE=1"; do
  echo "== $v"; env $v LVX_BENCH_NOPROF=1 python bench.py --no-secondary --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
