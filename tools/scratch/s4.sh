#!/bin/bash
# round 6, session 4: full-line stores + HW_ID roles variant against the round-5 library, then the GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
date
timeout 900 python tools/scratch/chk_const4.py 2>&1 | grep -E "RESULT|constant channels|max\|dprob" | sort | uniq -c | sort -rn | head -12
ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_hwid.so timeout 900 python tools/scratch/chk_const4.py 2>&1 | grep -E "RESULT|max\|dprob" | sort | uniq -c | sort -rn | head -8
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="new (full lines)"; line
V="hwid roles"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_hwid.so line
V="r5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line
done
date
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -25
date
} > gpurun_out/r6/s4.log 2>&1
tail -60 gpurun_out/r6/s4.log
