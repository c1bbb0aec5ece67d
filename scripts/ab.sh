#!/bin/bash
# One documented A/B runner (replaces the round-3 scripts/experiments/exp*.sh one-offs).
#
#   bash scripts/ab.sh [-c CONFIG] [-r ROUNDS] [-s STEPS] [-w WARMUP] [-x "extra bench.py flags"] ARM [ARM ...]
#
# Every ARM is a (possibly empty) string of environment assignments, e.g. "" "ADELIE_HIP_LIB=libadelie_hip_x.so"
# "ADELIE_HIP_PANEL_BSZ=64".  The arms are run ROUNDS times INTERLEAVED (arm1 arm2 ... arm1 arm2 ...) on the same box,
# because box-to-box spread (+-4 %) is larger than most effects being measured; one line per run:
#   cfg arm | paths/s ms/path | sweep cd gram host-compute host-wait (ms of the last path) | fused-launch us
# Run it through gpurun:  gpurun --timeout 900 -- 'bash scripts/ab.sh -c 2 -r 3 "" "ADELIE_HIP_LIB=libadelie_hip_v.so"'
CFG=2; ROUNDS=3; STEPS=5; WARM=2; EXTRA=""
while getopts "c:r:s:w:x:" o; do
  case $o in c) CFG=$OPTARG;; r) ROUNDS=$OPTARG;; s) STEPS=$OPTARG;; w) WARM=$OPTARG;; x) EXTRA=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs --config $CFG --steps $STEPS --warmup $WARM $EXTRA"
for ((i = 0; i < ROUNDS; ++i)); do
  for arm in "$@"; do
    env $arm $B 2>gpurun_out/ab_last.err | ARM="$arm" python -c "
import json, os, sys
lines = sys.stdin.read().strip().splitlines()
if not lines:
    print('cfg$CFG [%s] FAILED (see gpurun_out/ab_last.err)' % os.environ['ARM']); sys.exit(0)
d = json.loads(lines[-1])
b = d.get('breakdown_ms_last_path') or {}
ps = d.get('roofline_panel_step') or (d['roofline'] if 'panel' in (d.get('roofline') or {}).get('kernel', '') else {})
g = lambda k: round(b.get(k, float('nan')), 1)
print('cfg$CFG [%s] | %.4f %.1f | sweep %s cd %s gram %s hostc %s hostw %s | fused %.1f us' % (
    os.environ['ARM'], d['value'], d['ms_per_step'], g('sweep'), g('cd'), g('gram_mfma'), g('host_screen_compute'),
    g('host_screen_wait'), 1e3 * ps.get('avg_launch_ms', float('nan'))))
"
  done
done
