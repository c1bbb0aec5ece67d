#!/bin/bash
# Builds adelie_amd/libadelie_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# AHIP_VARIANT=name builds libadelie_hip_name.so from objects under build_name/ (compile-time A/B variants, with AHIP_EXTRA_FLAGS)
V="${AHIP_VARIANT:-}"
OUT="$HERE/../libadelie_hip${V:+_$V}.so"
OBJ="$HERE/build${V:+_$V}"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${AHIP_EXTRA_FLAGS:-}"
pids=()
for f in kernels_cons kernels_append kernels_eig kernels_strip kernels_cov kernels_sparse kernels_sweep kernels_gram kernels_cd kernels_cd_lasso kernels_cd_block kernels_cd_block_group kernels_cd_panel kernels_multi kernels_glm design solver; do
  src="$HERE/$f.hip"; obj="$OBJ/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/kernels.hpp" -nt "$obj" ] || [ "$HERE/common.hpp" -nt "$obj" ] \
     || [ "$HERE/accessors.hpp" -nt "$obj" ] || [ "$HERE/wavered.hpp" -nt "$obj" ] || [ "$HERE/gram_common.hpp" -nt "$obj" ] || [ "$HERE/blk_solve_body.hpp" -nt "$obj" ] || [ "$HERE/grp_solve_body.hpp" -nt "$obj" ] || { [ "$f" = solver ] && [ -n "$(find "$HERE" -name "solver_*.hpp" -newer "$obj" 2>/dev/null)" ]; } || [ "$HERE/../../include/adelie_hip.h" -nt "$obj" ]; then
    timeout 600 $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/kernels_cons.o "$OBJ"/kernels_append.o "$OBJ"/kernels_eig.o "$OBJ"/kernels_strip.o "$OBJ"/kernels_cov.o "$OBJ"/kernels_sparse.o "$OBJ"/kernels_sweep.o "$OBJ"/kernels_gram.o "$OBJ"/kernels_cd.o "$OBJ"/kernels_cd_lasso.o "$OBJ"/kernels_cd_block.o "$OBJ"/kernels_cd_block_group.o "$OBJ"/kernels_cd_panel.o "$OBJ"/kernels_multi.o \
  "$OBJ"/kernels_glm.o "$OBJ"/design.o "$OBJ"/solver.o
echo "built $OUT"
