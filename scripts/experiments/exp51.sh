cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
run() { env "$@" $B --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $*', round(d['value'],4), round(d['ms_per_step'],1), d['counters']['n_speculated'])"; }
run A=0
run ADELIE_HIP_LOOKAHEAD_MIN_BLOCKS=2
run ADELIE_HIP_CD_BLOCK_MIN_NV=64
run ADELIE_HIP_CD_BLOCK_MIN_NV=64 ADELIE_HIP_LOOKAHEAD_MIN_BLOCKS=2
run A=0
run ADELIE_HIP_CD_BLOCK_MIN_NV=32 ADELIE_HIP_LOOKAHEAD_MIN_BLOCKS=2
run ADELIE_HIP_CD_BLOCK_MIN_NV=96
