#!/bin/bash
# Experimental build of the rw stage kernel only: tools/build_variant.sh NAME [extra hipcc flags for rn_stage_rw.hip]
# -> roomnet_amd/lib/libroomnet_hip_NAME.so (other objects are taken from build/obj: run csrc/build.sh first).
# Select at run time with ROOMNET_HIP_LIB=<path>.  Diagnostic only; never shipped.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/roomnet_amd/csrc"
NAME="$1"; shift
OBJ="$ROOT/build/var_$NAME"; mkdir -p "$OBJ"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -fvisibility=hidden -I"$ROOT/include" -I"$SRC" \
    -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -DRN_BUILDING -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c "$SRC/rn_stage_rw.hip" -o "$OBJ/rn_stage_rw.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$ROOT"/build/obj/rn_api.o "$ROOT"/build/obj/rn_kernels_f32.o \
    "$ROOT"/build/obj/rn_fused.o "$ROOT"/build/obj/rn_imageops.o "$OBJ/rn_stage_rw.o" -o "$ROOT/roomnet_amd/lib/libroomnet_hip_$NAME.so"
echo "built libroomnet_hip_$NAME.so"
