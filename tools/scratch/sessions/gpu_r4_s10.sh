#!/bin/bash
# GPU box, round 4 session 10: back-end geometry test; timing split of the back-end launch (variant without its tail phase: wrong results)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_fused.py -m gpu -q -x -k "geometry_edges or back_end" 2>&1 | tail -8 | tee gpurun_out/r4/s10_pytest.txt
tools/gpu_var.sh base notail base notail 2>&1 | tee gpurun_out/r4/s10_ab.txt
