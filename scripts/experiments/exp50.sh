cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_golden.py tests/test_gpu_solver.py -m gpu -q -x 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for i in 1 2 3; do
$B --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['value'],4), round(d['ms_per_step'],1), round(d['breakdown_ms_last_path']['gram_mfma'],1), round(d['breakdown_ms_last_path']['cd'],1), round(d['roofline_gram_mfma']['hbm_gbs']))"
done
for i in 1 2; do
$B --config 3 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', round(d['value'],4), round(d['ms_per_step'],1), round(d['breakdown_ms_last_path']['gram_mfma'],1))"
done
