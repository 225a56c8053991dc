#!/bin/bash
# GPU box, round 4 session 11: back end with stage 7 split by kernel row: tests + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_fused.py -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r4/s11_pytest.txt
tools/gpu_var.sh r3 base r3 base 2>&1 | tee gpurun_out/r4/s11_ab.txt
