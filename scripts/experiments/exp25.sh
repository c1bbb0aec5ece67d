cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "group or grp or golden or constraint or panel or lookahead or lasso or gaussian" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for t in 1 1; do
ADELIE_HIP_TRACE_ENQ=1 $B --config 3 --steps 3 --warmup 1 2>gpurun_out/e25_$t.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
grep "enq\]" gpurun_out/e25_$t.err | tail -1
ADELIE_HIP_TRACE_ENQ=1 $B --config 2 --steps 5 --warmup 2 2>gpurun_out/e25b_$t.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()})"
grep "enq\]" gpurun_out/e25b_$t.err | tail -1
done
