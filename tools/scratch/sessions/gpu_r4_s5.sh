#!/bin/bash
# GPU box, round 4 session 5: conv16p operand prefetch depth, stage-5 transposed-read mapping, measurement ladder
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_k5.so python -m pytest tests/test_hip_fused.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r4/s5_pytest_k5.txt
ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_p6.so python -m pytest tests/test_hip_fused.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r4/s5_pytest_p6.txt
tools/gpu_var.sh base p2 p4 p6 p8 k5 base p2 p4 p6 p8 k5 2>&1 | tee gpurun_out/r4/s5_ab.txt
bash tools/gpu_ladder.sh
