#!/bin/bash
# Build libroomnet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/roomnet_amd/lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -fPIC -shared -fvisibility=hidden
       -I"$ROOT/include" -I"$HERE" -Wall -Wno-unused-function -DRN_BUILDING)
# objects are built separately so that per-file scheduler options can be applied
OBJ="$ROOT/build/obj"
mkdir -p "$OBJ"
CFLAGS=("${FLAGS[@]/-shared/}")
# (a failed background compile must fail the build: a bare `wait` returns 0 and the link would pick up a stale object)
PIDS=()
for f in rn_api rn_kernels_f32 rn_fused rn_imageops rn_group rn_tail rn_conv16 rn_stage_f32m rn_backend; do
    rm -f "$OBJ/$f.o"
    "$HIPCC" "${CFLAGS[@]}" -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    PIDS+=($!)
done
# MFMA results stay in VGPRs: the epilogue reads every accumulator with the VALU, and AGPR
# accumulators cost one v_accvgpr_read each (64 per row in the residual variant).
# (max-ilp scheduling was measured slower, see NOTES.md)
rm -f "$OBJ/rn_stage_rw.o"
"$HIPCC" "${CFLAGS[@]}" ${RN_RW_FLAGS:--mllvm -amdgpu-mfma-vgpr-form} -c "$HERE/rn_stage_rw.hip" -o "$OBJ/rn_stage_rw.o" &
PIDS+=($!)
rm -f "$OBJ/rn_stage23.o"
"$HIPCC" "${CFLAGS[@]}" ${RN_RW_FLAGS:--mllvm -amdgpu-mfma-vgpr-form} -c "$HERE/rn_stage23.hip" -o "$OBJ/rn_stage23.o" &
PIDS+=($!)
rm -f "$OBJ/rn_stage23x.o"
"$HIPCC" "${CFLAGS[@]}" ${RN_RW_FLAGS:--mllvm -amdgpu-mfma-vgpr-form} -c "$HERE/rn_stage23x.hip" -o "$OBJ/rn_stage23x.o" &
PIDS+=($!)
rm -f "$OBJ/rn_stage5x.o"
"$HIPCC" "${CFLAGS[@]}" ${RN_RW_FLAGS:--mllvm -amdgpu-mfma-vgpr-form} -c "$HERE/rn_stage5x.hip" -o "$OBJ/rn_stage5x.o" &
PIDS+=($!)
rm -f "$OBJ/rn_stage4x.o"
"$HIPCC" "${CFLAGS[@]}" ${RN_RW_FLAGS:--mllvm -amdgpu-mfma-vgpr-form} -c "$HERE/rn_stage4x.hip" -o "$OBJ/rn_stage4x.o" &
PIDS+=($!)
rm -f "$OBJ/rn_stage6x.o"
"$HIPCC" "${CFLAGS[@]}" ${RN_RW_FLAGS:--mllvm -amdgpu-mfma-vgpr-form} -c "$HERE/rn_stage6x.hip" -o "$OBJ/rn_stage6x.o" &
PIDS+=($!)
for p in "${PIDS[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OBJ"/rn_api.o "$OBJ"/rn_kernels_f32.o "$OBJ"/rn_fused.o "$OBJ"/rn_imageops.o "$OBJ"/rn_group.o "$OBJ"/rn_tail.o "$OBJ"/rn_conv16.o "$OBJ"/rn_stage_rw.o "$OBJ"/rn_stage23x.o "$OBJ"/rn_stage5x.o "$OBJ"/rn_stage4x.o "$OBJ"/rn_stage6x.o "$OBJ"/rn_stage_f32m.o "$OBJ"/rn_backend.o -ldl -lpthread \
    ${RN_EXTRA_FLAGS:-} -o "$OUT/libroomnet_hip.so"
echo "built $OUT/libroomnet_hip.so"
# the test / A-B library: the same objects + the round-2 comparison kernels (RN_FLAG_PAIR_32X32: rn_stage23.hip), which the
# product library does not carry
AB="$OBJ/ab"
mkdir -p "$AB"
rm -f "$AB"/rn_api.o "$AB"/rn_fused.o
"$HIPCC" "${CFLAGS[@]}" -DRN_ROUND2_ARMS -c "$HERE/rn_api.hip" -o "$AB/rn_api.o" &
P1=$!
"$HIPCC" "${CFLAGS[@]}" -DRN_ROUND2_ARMS -c "$HERE/rn_fused.hip" -o "$AB/rn_fused.o" &
P2=$!
wait "$P1"; wait "$P2"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$AB"/rn_api.o "$OBJ"/rn_kernels_f32.o "$AB"/rn_fused.o "$OBJ"/rn_imageops.o "$OBJ"/rn_group.o "$OBJ"/rn_tail.o "$OBJ"/rn_conv16.o "$OBJ"/rn_stage_rw.o "$OBJ"/rn_stage23.o "$OBJ"/rn_stage23x.o "$OBJ"/rn_stage5x.o "$OBJ"/rn_stage4x.o "$OBJ"/rn_stage6x.o "$OBJ"/rn_stage_f32m.o "$OBJ"/rn_backend.o -ldl -lpthread \
    -o "$OUT/libroomnet_hip_ab.so"
echo "built $OUT/libroomnet_hip_ab.so"
# register report of the hot kernels: a spill in one of them costs ~25 % of its time (seen on the fused stage pair) and
# hipcc does not warn about it
if [ -x "$ROOT/tools/spills.sh" ]; then
    "$ROOT/tools/spills.sh" "$OBJ"/rn_stage23.o "$OBJ"/rn_stage23x.o "$OBJ"/rn_stage5x.o "$OBJ"/rn_stage4x.o "$OBJ"/rn_stage6x.o "$OBJ"/rn_stage_rw.o "$OBJ"/rn_tail.o "$OBJ"/rn_conv16.o | awk '$0 ~ /spills +[1-9]/ {print "  spills: " $0}' | cut -c1-70,95-200 || true
fi
