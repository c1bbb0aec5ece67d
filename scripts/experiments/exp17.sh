cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "panel or lookahead or golden or specul or lasso or gaussian" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for mm in 128 64 128 0; do
ADELIE_HIP_TRACE_ENQ=1 ADELIE_HIP_STRIP_MAX_M=$mm ADELIE_HIP_STRIP_BUILDS=$((mm>0)) $B --config 2 --steps 5 --warmup 2 > gpurun_out/e17_cfg2_$mm.json 2>gpurun_out/e17_cfg2_$mm.err
python - <<PY
import json
d=json.loads(open("gpurun_out/e17_cfg2_$mm.json").read().strip().splitlines()[-1])
print("max_m=$mm", round(d["value"],4), round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["breakdown_ms_last_path"].items()})
PY
grep "enq\]" gpurun_out/e17_cfg2_$mm.err | tail -1
done
for mm in 128 0; do
ADELIE_HIP_STRIP_BUILDS=$((mm>0)) $B --config 2 --dtype f32 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 max_m=$mm', d['value'], d['ms_per_step'])"
done
for mm in 128 0 128 0; do
ADELIE_HIP_STRIP_BUILDS=$((mm>0)) $B --config 5 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 max_m=$mm', d['value'], d['ms_per_step'])"
done
