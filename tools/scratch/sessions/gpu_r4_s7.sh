#!/bin/bash
# GPU box, round 4 session 7: debug the float32 matrix-core path; the one-launch back end (tests + A/B)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python tools/scratch/dbg_f32m.py 2>&1 | tail -30 | tee gpurun_out/r4/s7_dbg_f32m.txt
python -m pytest tests/test_hip_fused.py -m gpu -q -x -k "back_end or batch_geometry or full_batch_256" 2>&1 | tail -15 | tee gpurun_out/r4/s7_pytest_backend.txt
tools/gpu_var.sh r3 base r3 base 2>&1 | tee gpurun_out/r4/s7_ab.txt
