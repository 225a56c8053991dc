#!/bin/bash
# GPU box: one library variant of the 16x16x32 fused pair: agreement with the 32x32x16 kernel + timing of both
cd $GRAFT_REPO_ROOT
export ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_$1.so
timeout 300 python tools/scratch/dbg_pairx.py 2>&1 | tail -8
tools/gpu_ab2.sh "" "--pair32"
