set -x
cd $GRAFT_REPO_ROOT
timeout 1700 python bench.py --steps 5 --warmup 2 > gpurun_out/e4_default.json 2> gpurun_out/e4_default.err; echo rc=$?
tail -c 600 gpurun_out/e4_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/e4_default.json").read().strip().splitlines()[-1])
def show(tag, x):
    print(tag, round(x["value"],4), x["unit"], round(x["ms_per_step"],1), "ms; roofline", x["roofline"]["kernel"][:28], round(x["roofline"]["frac"],3), "path", round(x["roofline_path"]["frac"],3), "panel", (round(x["roofline_panel_step"]["frac"],3) if x.get("roofline_panel_step") else None))
    print("   ", x.get("breakdown_ms_last_path"))
show("cfg2", d)
for k in ("cfg3","f32","cfg4"):
    if k in d: show(k, d[k])
print("cv", d.get("cv_config5"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","seconds","lambdas_solved","max_abs_dbeta_vs_gpu")})
PY
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/e4_cfg5.json 2> gpurun_out/e4_cfg5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/e4_cfg5.json").read().strip().splitlines()[-1])
print("cfg5", round(d["value"],3), round(d["ms_per_step"],1), "roofline", round(d["roofline"]["frac"],3), "path", round(d["roofline_path"]["frac"],3), d["roofline_path"]["terms_in_columns"])
PY
python scripts/py_profile.py 2>&1 | head -50
