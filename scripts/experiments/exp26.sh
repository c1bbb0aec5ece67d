cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "group or grp or golden or constraint or multi" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for t in 1 0 1 0; do
ADELIE_HIP_UV_SIDE=$t $B --config 3 --steps 3 --warmup 1 2>gpurun_out/e26_$t.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 uv_side=$t', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
done
