#!/bin/bash
# round 6, session 1: const4 fold correctness + first timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
timeout 900 python tools/scratch/chk_const4.py
echo "== bench (driver cmd)"
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3
echo "== bench 200"
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1
echo "== r5 lib for comparison"
} > gpurun_out/r6/s1.log 2>&1
tail -5 gpurun_out/r6/s1.log
