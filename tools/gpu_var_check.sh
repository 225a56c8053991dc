#!/bin/bash
# GPU box: time library variants side by side WITH the parity gate on (results must match the goldens): tools/gpu_var_check.sh name...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_$v.so; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.readline()
try:
    d=json.loads(l)
    print('%-12s %.0f img/s  ' % ('$v', d['value']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + '  parity ids_wrong %s dprob %.2g' % (d['parity'].get('ids_wrong'), d['parity'].get('max_abs_dprob', -1)))
except Exception as e:
    print('$v', 'FAILED', l[:300])"
done; done
