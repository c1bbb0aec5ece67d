cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_multi.py tests/test_cv.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 3 1 3 0 3; do
ADELIE_HIP_MULTI_SWEEP_WPC=$v $B --config 5 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5 wpc=$v', round(d['value'],4), round(d['ms_per_step'],1), round(r['avg_launch_ms'],4), round(r['frac'],3), r.get('vectors_per_launch'))"
done
