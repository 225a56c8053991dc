#!/bin/bash
# Incremental rebuild: tools/build_inc.sh rn_stage23x [rn_api ...]  (objects of the other files are reused from build/obj;
# run roomnet_amd/csrc/build.sh once first)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/roomnet_amd/csrc"; OBJ="$ROOT/build/obj"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -fvisibility=hidden -I"$ROOT/include" -I"$SRC" -Wall -Wno-unused-function -DRN_BUILDING)
ALL="rn_api rn_kernels_f32 rn_fused rn_imageops rn_group rn_tail rn_conv16 rn_stage_f32m rn_backend rn_stage_rw rn_stage23 rn_stage23x rn_stage5x rn_stage4x rn_stage6x"
PIDS=()
for f in "$@"; do
  extra=(); case $f in rn_stage_rw|rn_stage23|rn_stage23x|rn_stage5x|rn_stage4x|rn_stage6x) extra=(-mllvm -amdgpu-mfma-vgpr-form);; esac
  rm -f "$OBJ/$f.o"
  /opt/rocm/bin/hipcc "${FLAGS[@]}" "${extra[@]}" -c "$SRC/$f.hip" -o "$OBJ/$f.o" &
  PIDS+=($!)
done
for p in "${PIDS[@]}"; do wait "$p"; done
OBJS=(); for f in $ALL; do OBJS+=("$OBJ/$f.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -ldl -lpthread -o "$ROOT/roomnet_amd/lib/libroomnet_hip.so"
echo "built libroomnet_hip.so"
for f in "$@"; do "$ROOT/tools/spills.sh" "$OBJ/$f.o" | awk '$0 ~ /spills +[1-9]/ {print "  spills: " $0}' | cut -c1-70,95-200 || true; done
