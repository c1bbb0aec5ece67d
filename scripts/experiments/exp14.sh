cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs --steps 5 --warmup 2"
for v in 256 128 64 16; do
  ADELIE_HIP_CD_BLOCK_MIN_NV=$v $B > gpurun_out/e14_$v.json 2>gpurun_out/e14_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/e14_$v.json").read().strip().splitlines()[-1])
print("min_nv=$v", round(d["value"],3), round(d["ms_per_step"],1), d["counters"]["n_speculated"]//5)
PY
done
for v in 2 4; do
  ADELIE_HIP_LOOKAHEAD_MIN_BLOCKS=$v $B > gpurun_out/e14_la$v.json 2>gpurun_out/e14_la$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/e14_la$v.json").read().strip().splitlines()[-1])
print("la_min_blocks=$v", round(d["value"],3), round(d["ms_per_step"],1))
PY
done
