#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_fused.py -m gpu -q -x -k "stage_outputs or golden or geometry_edges or batch_geometry" 2>&1 | tail -4 | tee gpurun_out/r4/s15_pytest.txt
tools/gpu_var.sh pre base pre base pre base 2>&1 | tee gpurun_out/r4/s15_ab.txt
