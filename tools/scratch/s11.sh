#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
date
timeout 900 python tools/scratch/chk_const4.py 2>&1 | grep -E "RESULT|s3.bn2|dprob|Error|error" | head -40
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="new (8-channel ring)"; line
V="r5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line
done
V="new f16"; line --dtype f16
V="r5 f16"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line --dtype f16
V="new 600 f16"; line --dtype f16 --side 600 --batch 64
V="r5 600 f16"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line --dtype f16 --side 600 --batch 64
date
} > gpurun_out/r6/s11.log 2>&1
tail -60 gpurun_out/r6/s11.log | cut -c1-260
