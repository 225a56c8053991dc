#!/bin/bash
# Experimental build of ONE kernel file: tools/build_variant.sh NAME FILE [extra hipcc flags]   (FILE = rn_stage_rw | rn_stage23)
# -> tools/ab/libroomnet_hip_NAME.so (other objects are taken from build/obj: run csrc/build.sh first).
# Select at run time with ROOMNET_HIP_LIB=<path>.  Diagnostic only; never shipped.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/roomnet_amd/csrc"
NAME="$1"; FILE="$2"; shift 2
OBJ="$ROOT/build/var_$NAME"; mkdir -p "$OBJ"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -fvisibility=hidden -I"$ROOT/include" -I"$SRC" \
    -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -DRN_BUILDING -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c "$SRC/$FILE.hip" -o "$OBJ/$FILE.o"
OBJS=()
for f in rn_api rn_kernels_f32 rn_fused rn_imageops rn_group rn_tail rn_conv16 rn_stage_rw rn_stage23 rn_stage23x rn_stage5x rn_stage4x rn_stage6x rn_stage_f32m rn_backend; do
  if [ "$f" = "$FILE" ]; then OBJS+=("$OBJ/$f.o"); else OBJS+=("$ROOT/build/obj/$f.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -ldl -lpthread -o "$ROOT/tools/ab/libroomnet_hip_$NAME.so"
echo "built libroomnet_hip_$NAME.so"
"$ROOT/tools/spills.sh" "$OBJ/$FILE.o" | awk '$0 ~ /spills +[1-9]/ {print "  spills: " $0}' | cut -c1-70,95-200 || true
