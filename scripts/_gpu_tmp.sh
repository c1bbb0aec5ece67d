mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r02_xb8_pytest.log 2>&1; echo "full gpu rc=$?"; tail -2 gpurun_out/r02_xb8_pytest.log | cut -c1-200
run() { echo "== $*"; for i in 1 2 3; do env "$@" timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline --no-cv-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; }
run ADELIE_HIP_CROSS_BATCH=8
run ADELIE_HIP_CROSS_BATCH=16
run ADELIE_HIP_CROSS_BATCH=1
run ADELIE_HIP_CROSS_BATCH=16 ADELIE_HIP_BATCH_BLOCKS=16
