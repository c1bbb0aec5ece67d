mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_constraint.py tests/test_gpu_solver.py tests/test_multi.py -q > gpurun_out/r02_cons2_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02_cons2_pytest.log | cut -c1-200
