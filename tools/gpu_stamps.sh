#!/bin/bash
# GPU box: stage-2+3 stamps of stamp-build variants: tools/gpu_stamps.sh NAME...   (libroomnet_hip_stamps_NAME.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for v in "$@"; do
  echo "== $v"
  ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_stamps_$v.so python bench.py --steps 2 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-parity-check --profile-steps 1 2>&1 | grep -E "stamps" | grep -E "segment|wave 0|wave 4" | tail -5
done 2>&1 | tee gpurun_out/r3/stamps_multi.txt
