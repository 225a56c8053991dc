#!/bin/bash
# ISA report of one kernel source: instruction classes per barrier-delimited step, where the scratch (spill) instructions sit,
# packed-fp32 and AGPR-copy counts.  usage: tools/asm_report.sh FILE(.hip, in roomnet_amd/csrc) KERNEL_SUBSTRING [extra hipcc flags]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
F="$1"; K="$2"; shift 2
mkdir -p "$ROOT/build/asm"
S="$ROOT/build/asm/$F.s"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -I"$ROOT/include" -I"$ROOT/roomnet_amd/csrc" -DRN_BUILDING \
    -mllvm -amdgpu-mfma-vgpr-form "$@" -S --cuda-device-only "$ROOT/roomnet_amd/csrc/$F.hip" -o "$S" 2>/dev/null
python3 "$ROOT/tools/asm_steps.py" "$S" "$K"
awk -v k="$K" '$0 ~ /@function/ {f = index($0, k) > 0} f' "$S" > "$S.k"
echo "scratch / barrier lines: $(grep -n 'scratch_\|s_barrier' "$S.k" | awk '{printf "%s%s ", $1, substr($2,1,9)}')"
echo "v_pk_fma/mul/add_f32: $(grep -c 'v_pk_fma_f32\|v_pk_mul_f32\|v_pk_add_f32' "$S.k")  accvgpr moves: $(grep -c accvgpr "$S.k")"
