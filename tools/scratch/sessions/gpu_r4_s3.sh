#!/bin/bash
# GPU box, round 4 session 3: column-block row-blocked kernels (420 / 600), cleaned pair kernel (0 / 2 / 4 fragments in LDS), tail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4/s3_pytest.txt
tools/gpu_var.sh r3 c0 c2 base r3 c0 c2 base 2>&1 | tee gpurun_out/r4/s3_ab.txt
for v in r3 base; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_$v.so; fi
  for rep in 1 2; do
  python bench.py --side 600 --batch 64 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('600 %-6s %.0f img/s  cold %.0f  ' % ('$v', d['value'], d.get('cold_images_per_sec', 0)) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + ' head %.3f' % d['path']['head_ms'] + '  parity %s' % (d['parity'],))"
  done
done 2>&1 | tee gpurun_out/r4/s3_600.txt
