#!/bin/bash
# Register / spill report of the device code in hipcc object files: tools/spills.sh build/obj/rn_stage23.o [...]
# (llvm-objdump on the bundled host object shows no device code at all -- a "0 scratch instructions" read from it is meaningless)
set -euo pipefail
LLVM=/opt/rocm/lib/llvm/bin
for o in "$@"; do
  t=$(mktemp)
  f=$(mktemp)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin="$f" "$o" /dev/null
  $LLVM/clang-offload-bundler --unbundle --type=o --input="$f" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$t"
  rm -f "$f"
  echo "== $o"
  $LLVM/llvm-readelf --notes "$t" | awk '
    /\.name:/ {name=$2}
    /\.vgpr_count:/ {v=$2}
    /\.agpr_count:/ {a=$3}
    /\.vgpr_spill_count:/ {s=$2}
    /\.private_segment_fixed_size:/ {p=$2}
    /\.wavefront_size:/ {printf "%-90s vgpr %3s agpr %3s spills %3s scratch %4s B\n", substr(name,1,90), v, a, s, p}'
  rm -f "$t"
done
