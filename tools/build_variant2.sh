#!/bin/bash
# Experimental build with extra flags applied to SEVERAL kernel files: tools/build_variant2.sh NAME "FILE1 FILE2" [flags]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/roomnet_amd/csrc"
NAME="$1"; FILES="$2"; shift 2
OBJ="$ROOT/build/var_$NAME"; mkdir -p "$OBJ"
PIDS=()
for FILE in $FILES; do
  EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
  rm -f "$OBJ/$FILE.o"
  case "$FILE" in rn_api|rn_kernels_f32|rn_fused|rn_imageops|rn_group|rn_tail|rn_conv16) EXTRA="";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -fvisibility=hidden -I"$ROOT/include" -I"$SRC" \
      -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -DRN_BUILDING $EXTRA "$@" \
      -c "$SRC/$FILE.hip" -o "$OBJ/$FILE.o" &
  PIDS+=($!)
done
for p in "${PIDS[@]}"; do wait "$p"; done
OBJS=()
for f in rn_api rn_kernels_f32 rn_fused rn_imageops rn_group rn_tail rn_conv16 rn_stage_rw rn_stage23 rn_stage23x rn_stage5x rn_stage4x rn_stage6x rn_stage_f32m rn_backend; do
  if [[ " $FILES " == *" $f "* ]]; then OBJS+=("$OBJ/$f.o"); else OBJS+=("$ROOT/build/obj/$f.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -ldl -lpthread -o "$ROOT/tools/ab/libroomnet_hip_$NAME.so"
echo "built libroomnet_hip_$NAME.so"
for FILE in $FILES; do "$ROOT/tools/spills.sh" "$OBJ/$FILE.o" | awk '$0 ~ /spills +[1-9]/ {print "  spills: " $0}' | cut -c1-70,95-200 || true; done
