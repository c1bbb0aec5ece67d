mkdir -p gpurun_out
bash scripts/prof_cmd.sh r02i 3 > gpurun_out/prof_r02i_cfg3.log 2>&1; head -14 gpurun_out/r02i_cfg3_rocprof_summary.txt | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02i_pytest.log | cut -c1-200
