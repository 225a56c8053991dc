#!/bin/bash
# GPU box: the stage 0 + 1 launch under timing-only variants (is it bound by its output stream?)   usage: tools/gpu_s01.sh v1 v2 ...
cd $GRAFT_REPO_ROOT
line() { python bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --no-parity-check --no-cold-pass "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-10s batch %3d  %.0f img/s  launches ' % ('$V', d['config']['images_per_gpu'], d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for V in "$@"; do
  if [ "$V" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_$V.so; fi
  line
done
