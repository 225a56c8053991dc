#!/bin/bash
# round 6, session 2: the f16 anomaly + same-box A/B of the round-5 library against the const4 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
timeout 600 python tools/scratch/chk_anom.py
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="new"; line
V="r5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line
done
for rep in 1 2; do
V="new f16"; line --dtype f16
V="r5 library f16"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line --dtype f16
V="new 600 f16"; line --dtype f16 --side 600 --batch 64
V="r5 600 f16"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line --dtype f16 --side 600 --batch 64
done
} > gpurun_out/r6/s2.log 2>&1
tail -30 gpurun_out/r6/s2.log
