cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 8 4; do
ADELIE_HIP_SWEEP_CB=$v $B --config 4 --steps 1 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 sweep_cb=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()}, d['roofline_sweep'] if 'roofline_sweep' in d else '')"
done
