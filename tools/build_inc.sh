#!/bin/bash
# Incremental rebuild: tools/build_inc.sh rn_stage23 [rn_stage_rw ...]  (objects of the other files are reused from build/obj)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/roomnet_amd/csrc"; OBJ="$ROOT/build/obj"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -fvisibility=hidden -I"$ROOT/include" -I"$SRC" -Wall -Wno-unused-function -DRN_BUILDING)
for f in "$@"; do
  extra=(); case $f in rn_stage_rw|rn_stage23) extra=(-mllvm -amdgpu-mfma-vgpr-form);; esac
  /opt/rocm/bin/hipcc "${FLAGS[@]}" "${extra[@]}" -c "$SRC/$f.hip" -o "$OBJ/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OBJ"/rn_api.o "$OBJ"/rn_kernels_f32.o "$OBJ"/rn_fused.o "$OBJ"/rn_imageops.o "$OBJ"/rn_group.o "$OBJ"/rn_tail.o "$OBJ"/rn_conv16.o \
  "$OBJ"/rn_stage_rw.o "$OBJ"/rn_stage23.o -ldl -o "$ROOT/roomnet_amd/lib/libroomnet_hip.so"
echo "built libroomnet_hip.so"
for f in "$@"; do "$ROOT/tools/spills.sh" "$OBJ/$f.o" | awk '$0 ~ /spills +[1-9]/ {print "  spills: " $0}' | cut -c1-70,95-200 || true; done
