mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_constraint.py tests/test_abi.py -q > gpurun_out/r02_cons_pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r02_cons_pytest.log | cut -c1-200
