#!/bin/bash
# GPU box: parity of the fused kernel + A/B timing (fused vs stage launches) + stamps of the fused kernel
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_fused.py -x -q -m gpu -k "cross_stage or stage_outputs" 2>&1 | tail -3
for arm in "" "--stage-launches"; do python bench.py --steps 50 --warmup 10 --no-cpu-baseline $arm 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-16s %.0f img/s  ' % ('$arm', d['value']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']))"; done
if [ -f tools/ab/libroomnet_hip_stamps.so ]; then
ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_stamps.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 2>&1 | grep -E "stamps. (fused|  wave 0)" | tail -2
fi
