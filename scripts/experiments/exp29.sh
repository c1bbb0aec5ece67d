cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "snp or binomial or glm or irls" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 16 4 16; do
ADELIE_HIP_SNP_STEP_VEC=$v $B --config 4 --steps 1 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 vec=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
done
