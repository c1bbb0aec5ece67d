mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_multi.py -m gpu -q -x > gpurun_out/r02_hooks_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_hooks_pytest.log
for i in 1 2; do timeout 300 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | cut -c1-140 | tail -1; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
MAPS_NAME=maps_a.txt timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dbg_a -o k -- python $R/scripts/dbg_maps.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-cv-leg > $R/gpurun_out/dbg_a.out 2> $R/gpurun_out/dbg_a.err; echo "A (default) rc=$?"
ADELIE_HIP_SWEEP_BATCH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dbg_b -o k -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-cv-leg > $R/gpurun_out/dbg_b.out 2> $R/gpurun_out/dbg_b.err; echo "B (no batching) rc=$?"
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/dbg_c -o k -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-cv-leg > $R/gpurun_out/dbg_c.out 2> $R/gpurun_out/dbg_c.err; echo "C (no --stats) rc=$?"
cd $R; rm -rf gpurun_out/dbg_a gpurun_out/dbg_b gpurun_out/dbg_c
