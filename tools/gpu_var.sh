#!/bin/bash
# GPU box: time library variants side by side: tools/gpu_var.sh name1 name2 ...  ("base" = shipped library; "stamps" prints stamps)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_$v.so; fi
  if [ "$v" = stamps ]; then
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 2>&1 | grep -E "stamps. (fused|  wave 0|  per job)" | tail -3
  else
    python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-12s %.0f img/s  ' % ('$v', d['value']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']))"
  fi
done
