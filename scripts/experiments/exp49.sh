cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 512 160 192 512 160 192 512 160 192; do
ADELIE_HIP_STRIP_WGS=$v $B --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 strip_wgs=$v', round(d['value'],4), round(d['ms_per_step'],1), round(d['breakdown_ms_last_path']['gram_mfma'],1), round(d['breakdown_ms_last_path']['cd'],1))"
done
for v in 512 160 192 512 160 192; do
ADELIE_HIP_STRIP_WGS=$v $B --config 3 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 strip_wgs=$v', round(d['value'],4), round(d['ms_per_step'],1))"
done
