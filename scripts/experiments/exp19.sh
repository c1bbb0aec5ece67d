cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
run() { # label, env...
  lab=$1; shift
  env "$@" $B --config 2 --steps 5 --warmup 2 > gpurun_out/e19_$lab.json 2>gpurun_out/e19_$lab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/e19_$lab.json").read().strip().splitlines()[-1])
print("$lab", round(d["value"],4), round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["breakdown_ms_last_path"].items()})
PY
}
run w56 ADELIE_HIP_STRIP_WGS=56
run w112 ADELIE_HIP_STRIP_WGS=112
run w168 ADELIE_HIP_STRIP_WGS=168
run w512 ADELIE_HIP_STRIP_WGS=512
run noside ADELIE_HIP_SIDE_GRAMS=0
run w56b ADELIE_HIP_STRIP_WGS=56
