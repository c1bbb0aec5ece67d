mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_solver.py tests/test_multi.py tests/test_glm_families.py -m gpu -q -x > gpurun_out/r02_desc_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_desc_pytest.log | cut -c1-200
for i in 1 2; do timeout 300 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | cut -c1-140 | tail -1; done
timeout 300 python scripts/bench_multi.py 2>&1 | tail -4
touch adelie_amd/csrc/grp_solve_body.hpp
AHIP_EXTRA_FLAGS=-DAHIP_GRP_PROFILE bash adelie_amd/csrc/build.sh 2>&1 | grep -E "error|built"
timeout 600 python scripts/grp_profile.py 2>&1 | tail -7
