# End-of-round artefacts: full -m gpu suite, one bench line per config (with CPU baselines), rocprof summaries.
TAG=${1:-r03}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_final_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_final_pytest.log
tail -3 gpurun_out/${TAG}_final_pytest.log
for c in 2 3 5 4; do
  timeout 900 python bench.py --config $c > gpurun_out/${TAG}_final_cfg$c.json 2> gpurun_out/${TAG}_final_cfg$c.err; echo "cfg $c rc=$?"
  cut -c1-260 gpurun_out/${TAG}_final_cfg$c.json
done
for c in 2 3 5 4; do bash scripts/prof_cmd.sh ${TAG}f $c > gpurun_out/prof_${TAG}f_cfg$c.log 2>&1; done
ls gpurun_out | grep ${TAG}f | head -20
