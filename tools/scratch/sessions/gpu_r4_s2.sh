#!/bin/bash
# GPU box, round 4 session 2: pair-kernel variants (quad sums; 0 / 2 / 4 weight fragments in LDS) and the tail with hoisted loads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_q5n4.so python -m pytest tests/test_hip_fused.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4/s2_pytest_q5n4.txt
ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_t1.so python -m pytest tests/test_hip_fused.py tests/test_roomnet_api.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4/s2_pytest_t1.txt
tools/gpu_var.sh r3 q5n0 q5n2 q5n4 t1 r3 q5n0 q5n2 q5n4 t1 2>&1 | tee gpurun_out/r4/s2_ab.txt
for v in r3 q5n4; do
  export ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_$v.so
  for dt in bf16 f16; do
    python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --dtype $dt 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('sustained %-6s %-5s %.0f img/s  cold %.0f  ' % ('$v', '$dt', d['value'], d.get('cold_images_per_sec', 0)) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + '  parity %s dprob %.4f' % (d['parity'].get('ids_wrong'), d['parity'].get('max_abs_dprob', -1)))"
  done
done 2>&1 | tee gpurun_out/r4/s2_sustained.txt
