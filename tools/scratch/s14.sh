#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
date
timeout 900 python tools/scratch/dbg600.py 2>&1 | grep -v f16
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-26s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="round 6 (dither 3,4,5)"; line
V="round 6 --compute-frozen"; line --compute-frozen
V="round-5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line
V="round-5 --compute-frozen"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line --compute-frozen
done
date
} > gpurun_out/r6/s14.log 2>&1
cat gpurun_out/r6/s14.log | cut -c1-250
