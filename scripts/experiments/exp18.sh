cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "panel or lookahead or golden or specul or lasso or gaussian" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
run() { # label, env...
  lab=$1; shift
  env "$@" $B --config 2 --steps 5 --warmup 2 > gpurun_out/e18_$lab.json 2>gpurun_out/e18_$lab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/e18_$lab.json").read().strip().splitlines()[-1])
print("$lab", round(d["value"],4), round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["breakdown_ms_last_path"].items()})
PY
}
run lds1 ADELIE_HIP_STRIP_LDS=1
run lds0 ADELIE_HIP_STRIP_LDS=0
run lds1_w256 ADELIE_HIP_STRIP_LDS=1 ADELIE_HIP_STRIP_WGS=256
run lds1_w384 ADELIE_HIP_STRIP_LDS=1 ADELIE_HIP_STRIP_WGS=384
run lds1_w768 ADELIE_HIP_STRIP_LDS=1 ADELIE_HIP_STRIP_WGS=768
run lds1b ADELIE_HIP_STRIP_LDS=1
for w in 512 256; do
ADELIE_HIP_STRIP_WGS=$w $B --config 5 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 wgs=$w', d['value'], d['ms_per_step'])"
done
