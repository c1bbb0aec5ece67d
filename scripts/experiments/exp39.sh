cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for i in 1 2; do
$B --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
done
for c in 3 5 4; do
st=2; [ $c = 4 ] && st=1
$B --config $c --steps $st --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
done
