cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
ADELIE_HIP_STEP_ROWS=100 timeout 600 python -m pytest tests/test_golden.py tests/test_gpu_solver.py -m gpu -x -q 2>&1 | tail -2
for v in 0 100 104 112 0 100; do
ADELIE_HIP_STEP_ROWS=$v $B --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 rows=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()}, round(d['roofline_panel_step']['avg_launch_ms']*1e3,2))"
done
for v in 0 100; do
ADELIE_HIP_SIDE_GRAMS=0 ADELIE_HIP_STEP_ROWS=$v $B --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 side=0 rows=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()}, round(d['roofline_panel_step']['avg_launch_ms']*1e3,2))"
done
