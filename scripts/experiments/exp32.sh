cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for i in 1 2; do
$B --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
done
ADELIE_HIP_TRACE_ENQ=1 $B --steps 2 --warmup 1 2>&1 >/dev/null | grep alloc | tail -2
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_solver.py -m gpu -x -q 2>&1 | tail -3
