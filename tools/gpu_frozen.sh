#!/bin/bash
# GPU box: frozen-channel folding on / off side by side (same library, RN_FLAG_COMPUTE_FROZEN is the "off" arm) and the round-4 library
cd $GRAFT_REPO_ROOT
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2; do
V="folded"; line "$@"
V="compute-frozen"; line --compute-frozen "$@"
V="r4 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r4.so line "$@"
done
