cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl_k
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_k -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cv-leg --no-extra-legs > $R/gpurun_out/tl_bench.json 2> $R/gpurun_out/tl.err
cd $R
python scripts/timeline.py $(find gpurun_out/tl_k -name "*results.db" | head -1) --top 30 > gpurun_out/tl_timeline.txt 2>&1
rm -rf gpurun_out/tl_k
ADELIE_HIP_TRACE_ENQ=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cv-leg --no-extra-legs 2> gpurun_out/tl_enq.err | cut -c1-300
python scripts/py_profile.py > gpurun_out/tl_pyprof.txt 2>&1
tail -5 gpurun_out/tl_timeline.txt
