cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "panel or lookahead or golden or specul or lasso or gaussian or snp" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
run() { # label, env...
  lab=$1; shift
  env "$@" $B --config 2 --steps 5 --warmup 2 > gpurun_out/e20_$lab.json 2>gpurun_out/e20_$lab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/e20_$lab.json").read().strip().splitlines()[-1])
print("$lab", round(d["value"],4), round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["breakdown_ms_last_path"].items()}, d.get("parity_vs_cpu"))
PY
}
run open1 ADELIE_HIP_LA_FUSED_OPEN=1
run open0 ADELIE_HIP_LA_FUSED_OPEN=0
run open1b ADELIE_HIP_LA_FUSED_OPEN=1
run open0b ADELIE_HIP_LA_FUSED_OPEN=0
for o in 1 0; do
ADELIE_HIP_LA_FUSED_OPEN=$o $B --config 2 --dtype f32 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 open=$o', d['value'], d['ms_per_step'])"
ADELIE_HIP_LA_FUSED_OPEN=$o $B --config 5 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 open=$o', d['value'], d['ms_per_step'])"
done
