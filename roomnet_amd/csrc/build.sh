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
"$HIPCC" "${FLAGS[@]}" "$HERE"/rn_api.hip "$HERE"/rn_kernels_f32.hip "$HERE"/rn_fused.hip "$HERE"/rn_stage_rw.hip \
    ${RN_EXTRA_FLAGS:-} -o "$OUT/libroomnet_hip.so"
echo "built $OUT/libroomnet_hip.so"
