run() { echo "== $*"; for i in 1 2; do env "$@" timeout 300 python bench.py --config 2 --steps 5 --warmup 1 --no-cpu-baseline --no-cv-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; }
run A=0
run ADELIE_HIP_SIDE_CU_RESERVE=192 ADELIE_HIP_SIDE_CU_PATTERN=0
run ADELIE_HIP_SIDE_CU_RESERVE=224 ADELIE_HIP_SIDE_CU_PATTERN=0
run ADELIE_HIP_SIDE_CU_RESERVE=128 ADELIE_HIP_SIDE_CU_PATTERN=0
