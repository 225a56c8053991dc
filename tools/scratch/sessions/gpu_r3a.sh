#!/bin/bash
# GPU box, round 3: parity of a stage-2+3 kernel variant + A/B timing against the shipped library + its stamps
# usage: tools/gpu_r3a.sh VARIANT [more variants...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
V="$1"
ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_$V.so timeout 900 python -m pytest tests/test_hip_fused.py -x -q -m gpu > gpurun_out/r3/pytest_$V.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3/pytest_$V.log
tools/gpu_var.sh base "$@" base "$@" | tee gpurun_out/r3/ab_$V.txt
if [ -f roomnet_amd/lib/libroomnet_hip_stamps.so ]; then
ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_stamps.so python bench.py --steps 2 --warmup 1 --spinup-steps 0 --no-cpu-baseline --profile-steps 1 2>&1 | grep -E "stamps" | grep -v "stage [145]" | tail -12 | tee gpurun_out/r3/stamps_$V.txt
fi
