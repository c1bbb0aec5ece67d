set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
$B --config 3 --steps 3 --warmup 1 > gpurun_out/e5_cfg3.json 2>gpurun_out/e5_cfg3.err
ADELIE_HIP_DEVICE_EIG=0 $B --config 3 --steps 3 --warmup 1 > gpurun_out/e5_cfg3_hosteig.json 2>gpurun_out/e5_cfg3_hosteig.err
$B --config 2 --steps 5 --warmup 2 > gpurun_out/e5_cfg2.json 2>gpurun_out/e5_cfg2.err
$B --config 5 --steps 3 --warmup 1 > gpurun_out/e5_cfg5.json 2>gpurun_out/e5_cfg5.err
for f in cfg3 cfg3_hosteig cfg2 cfg5; do python - <<PY
import json
d=json.loads(open("gpurun_out/e5_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],3), round(d["ms_per_step"],1), d["breakdown_ms_last_path"])
PY
done
