# End-of-round artefacts: full -m gpu suite, the default bench line, one bench line per config (with CPU baselines and the
# parity fields), rocprof summaries (kernel trace + PMC passes per config, constrained path, sparse path).
TAG=${1:-r05}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_gpu_pytest.log
tail -3 gpurun_out/${TAG}_gpu_pytest.log
timeout 1200 python bench.py > gpurun_out/${TAG}_default_line.json 2> gpurun_out/${TAG}_default_line.err; echo "default line rc=$?"
cut -c1-200 gpurun_out/${TAG}_default_line.json
for c in 2 3 5 4; do
  timeout 1200 python bench.py --config $c > gpurun_out/${TAG}_cfg${c}_bench.json 2> gpurun_out/${TAG}_cfg${c}_bench.err; echo "cfg $c rc=$?"
  cut -c1-200 gpurun_out/${TAG}_cfg${c}_bench.json
done
for c in 2 3 5 4; do bash scripts/prof_cmd.sh ${TAG} $c > gpurun_out/prof_${TAG}_cfg$c.log 2>&1; done
bash scripts/prof_cons.sh ${TAG} > /dev/null 2>&1
ls gpurun_out | grep ${TAG} | head -40
