cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 1 0 1 0; do
ADELIE_HIP_SIDE_GRAMS=$v $B --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 side=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()}, round(d['roofline_panel_step']['avg_launch_ms']*1e3,2))"
done
for v in 1 0; do
ADELIE_HIP_SIDE_GRAMS=$v $B --config 3 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 side=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
done
