#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
tools/ubench/cvt_sr
timeout 900 python -m pytest tests/test_hip_fused.py -m gpu -q -k "one_launch_back_end_is_bit or frozen_channels_fold_against" 2>&1 | grep -E "^E|Error|assert|passed|failed" | head -60
} > gpurun_out/r6/s6.log 2>&1
tail -80 gpurun_out/r6/s6.log
