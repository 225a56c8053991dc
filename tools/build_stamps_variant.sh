#!/bin/bash
# Stamp build (tools/build_stamps.sh must have run) with ONE file rebuilt with extra flags:
#   tools/build_stamps_variant.sh NAME FILE [-D...]  ->  tools/ab/libroomnet_hip_stamps_NAME.so   (diagnostic only)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/roomnet_amd/csrc"
NAME="$1"; FILE="$2"; shift 2
OBJ="$ROOT/build/stamps_$NAME"; mkdir -p "$OBJ"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -fvisibility=hidden -I"$ROOT/include" -I"$SRC" \
    -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -DRN_BUILDING -DRN_STAMPS -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c "$SRC/$FILE.hip" -o "$OBJ/$FILE.o"
OBJS=()
for f in rn_api rn_kernels_f32 rn_fused rn_imageops rn_group rn_tail rn_conv16 rn_stage_rw rn_stage23 rn_stage23x rn_stage5x rn_stage4x rn_stage6x rn_stage_f32m rn_backend; do
  if [ "$f" = "$FILE" ]; then OBJS+=("$OBJ/$f.o"); else OBJS+=("$ROOT/build/stamps/$f.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -ldl -lpthread -o "$ROOT/tools/ab/libroomnet_hip_stamps_$NAME.so"
echo "built libroomnet_hip_stamps_$NAME.so"
