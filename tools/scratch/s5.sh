#!/bin/bash
# round 6, session 5: failing tests in detail, batched resize (test + timing)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
timeout 900 python -m pytest tests/test_hip_fused.py -m gpu -q -k "one_launch_back_end_is_bit or frozen_channels_fold_against or table_value_at_every or reproducible_at_batch_160" 2>&1 | grep -v "^$" | tail -120
timeout 600 python -m pytest tests/test_hip_imageops.py -m gpu -q 2>&1 | tail -5
timeout 600 python tools/profile_imageops.py
} > gpurun_out/r6/s5.log 2>&1
tail -150 gpurun_out/r6/s5.log
