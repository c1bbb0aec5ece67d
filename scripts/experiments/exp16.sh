cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "panel or lookahead or golden or specul or lasso or gaussian" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for sb in 1 0 1 0; do
ADELIE_HIP_TRACE_ENQ=1 ADELIE_HIP_STRIP_BUILDS=$sb $B --config 2 --steps 5 --warmup 2 > gpurun_out/e16_cfg2_$sb.json 2>gpurun_out/e16_cfg2_$sb.err
python - <<PY
import json
d=json.loads(open("gpurun_out/e16_cfg2_$sb.json").read().strip().splitlines()[-1])
print("strips=$sb", round(d["value"],4), round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["breakdown_ms_last_path"].items()})
PY
grep "enq\]" gpurun_out/e16_cfg2_$sb.err | tail -1
done
ADELIE_HIP_STRIP_BUILDS=1 $B --config 2 --dtype f32 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 strips=1', d['value'], d['ms_per_step'])"
ADELIE_HIP_STRIP_BUILDS=0 $B --config 2 --dtype f32 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 strips=0', d['value'], d['ms_per_step'])"
for sb in 1 0; do
ADELIE_HIP_STRIP_BUILDS=$sb $B --config 5 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 strips=$sb', d['value'], d['ms_per_step'])"
done
