cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 1 0 1; do
ADELIE_HIP_SIDE_GRAMS=$v $B --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 side=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()}, round(d['roofline_panel_step']['avg_launch_ms']*1e3,2))"
done
$B --config 3 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
$B --dtype f32 --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32', round(d['value'],4), round(d['ms_per_step'],1))"
