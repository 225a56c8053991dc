#!/bin/bash
# GPU box: library variants side by side at 64 x 600x600 fp16: tools/gpu_var600.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_$v.so; fi
  python bench.py --side 600 --batch 64 --dtype f16 --steps 60 --warmup 5 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-12s %.0f img/s  ' % ('$v', d['value']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']))"
done
