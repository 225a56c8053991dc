#!/bin/bash
# usage (on the GPU box): tools/run_variants.sh name1 name2 ...   ("base" = the shipped library)
for v in "$@"; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --profile-steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-12s %8.0f img/s  stage_ms %s head %.3f' % ('$v', d['value'], ' '.join('%.3f' % x for x in d['path']['stage_ms'][:8]), d['path']['head_ms']))"
done
