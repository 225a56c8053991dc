#!/bin/bash
# Diagnostic build with in-kernel clock stamps (never shipped): tools/ab/libroomnet_hip_clock.so
# Every stage-kernel workgroup stamps s_memtime / s_memrealtime at entry and exit; rn_forward prints the median clock per launch.
# Run AFTER >= 2 s of back-to-back launches (tools/gpu_clock.sh does that).  (run csrc/build.sh first: other objects are reused)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
exec "$ROOT/tools/build_variant2.sh" clock "rn_fused rn_stage_rw rn_stage23x rn_stage4x rn_stage5x rn_stage6x" -DRN_CLOCK
