#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never shipped): tools/ab/libroomnet_hip_stamps[_chain].so
# usage: tools/build_stamps.sh [chain]     then run with ROOMNET_HIP_LIB=<that .so> python bench.py --steps 1 --warmup 1 --no-cpu-baseline
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/roomnet_amd/csrc"
SUF=""; DEFS=(-DRN_STAMPS)
if [ "${1:-}" = "chain" ]; then SUF="_chain"; DEFS+=(-DRN_STAMP_CHAIN); fi
if [ "${1:-}" = "hwid" ]; then SUF="_hwid"; DEFS+=(-DRN_STAMP_HWID); fi
OBJ="$ROOT/build/stamps$SUF"; mkdir -p "$OBJ"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -fvisibility=hidden -I"$ROOT/include" -I"$SRC"
       -Wall -Wno-unused-function -DRN_BUILDING "${DEFS[@]}")
for f in rn_api rn_kernels_f32 rn_fused rn_imageops rn_group rn_tail rn_conv16 rn_stage_f32m rn_backend; do /opt/rocm/bin/hipcc "${FLAGS[@]}" -c "$SRC/$f.hip" -o "$OBJ/$f.o" & done
/opt/rocm/bin/hipcc "${FLAGS[@]}" -mllvm -amdgpu-mfma-vgpr-form -c "$SRC/rn_stage_rw.hip" -o "$OBJ/rn_stage_rw.o" &
/opt/rocm/bin/hipcc "${FLAGS[@]}" -mllvm -amdgpu-mfma-vgpr-form -c "$SRC/rn_stage23.hip" -o "$OBJ/rn_stage23.o" &
/opt/rocm/bin/hipcc "${FLAGS[@]}" -mllvm -amdgpu-mfma-vgpr-form -c "$SRC/rn_stage23x.hip" -o "$OBJ/rn_stage23x.o" &
/opt/rocm/bin/hipcc "${FLAGS[@]}" -mllvm -amdgpu-mfma-vgpr-form -c "$SRC/rn_stage5x.hip" -o "$OBJ/rn_stage5x.o" &
/opt/rocm/bin/hipcc "${FLAGS[@]}" -mllvm -amdgpu-mfma-vgpr-form -c "$SRC/rn_stage4x.hip" -o "$OBJ/rn_stage4x.o" &
/opt/rocm/bin/hipcc "${FLAGS[@]}" -mllvm -amdgpu-mfma-vgpr-form -c "$SRC/rn_stage6x.hip" -o "$OBJ/rn_stage6x.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -ldl -lpthread -o "$ROOT/tools/ab/libroomnet_hip_stamps$SUF.so"
echo "built $ROOT/tools/ab/libroomnet_hip_stamps$SUF.so"
