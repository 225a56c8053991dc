#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_pkepi.so timeout 900 python -m pytest tests/test_hip_fused.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-26s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="round 6"; line
V="pk epilogue"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_pkepi.so line
V="round 6 compute-frozen"; line --compute-frozen
V="pk epilogue compute-frozen"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_pkepi.so line --compute-frozen
done
} > gpurun_out/r6/s15.log 2>&1
cat gpurun_out/r6/s15.log | cut -c1-200
