mkdir -p gpurun_out
bash scripts/prof_cmd.sh r02h 5 > gpurun_out/prof_r02h_cfg5.log 2>&1; grep "attempt" gpurun_out/prof_r02h_cfg5.log; head -14 gpurun_out/r02h_cfg5_rocprof_summary.txt | cut -c1-150
bash scripts/prof_cmd.sh r02h 3 > gpurun_out/prof_r02h_cfg3.log 2>&1; grep "attempt" gpurun_out/prof_r02h_cfg3.log; head -12 gpurun_out/r02h_cfg3_rocprof_summary.txt | cut -c1-150
