#!/bin/bash
# GPU box: A/B of bench.py option sets inside one session: tools/gpu_ab2.sh "" "--pair32" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for rep in 1 2; do
for arm in "$@"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline $arm 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-28s %.0f img/s  ' % ('[$arm]', d['value']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + '  parity %s' % d['parity'].get('ids_wrong'))"
done; done 2>&1 | tee gpurun_out/r3/ab2.txt
