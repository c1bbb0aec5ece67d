cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
$B --config 4 --steps 1 --warmup 1 > gpurun_out/e15_cfg4.json 2>gpurun_out/e15_cfg4.err
ADELIE_HIP_SOLVE_WIDE=0 $B --config 4 --steps 1 --warmup 1 > gpurun_out/e15_cfg4_old.json 2>gpurun_out/e15_cfg4_old.err
$B --config 2 --steps 5 --warmup 2 > gpurun_out/e15_cfg2.json 2>gpurun_out/e15_cfg2.err
$B --config 5 --steps 3 --warmup 1 > gpurun_out/e15_cfg5.json 2>gpurun_out/e15_cfg5.err
for f in cfg4 cfg4_old cfg2 cfg5; do python - <<PY
import json
d=json.loads(open("gpurun_out/e15_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],4), round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["breakdown_ms_last_path"].items()})
PY
done
python scripts/bench_binom_dense.py 100000 10000 2>&1 | tail -2
