#!/bin/bash
# GPU box, round 4 session 1: GPU tests on the quad-sum library, then A/B against the round-3 library (short and sustained)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4/s1_pytest.txt
tools/gpu_var.sh r3 base r3 base 2>&1 | tee gpurun_out/r4/s1_ab.txt
for v in r3 base; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_$v.so; fi
  for dt in bf16 f16; do
    python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --dtype $dt 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('sustained %-6s %-5s %.0f img/s  ' % ('$v', '$dt', d['value']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + '  parity %s dprob %.4f' % (d['parity'].get('ids_wrong'), d['parity'].get('max_abs_dprob', -1)))"
  done
done 2>&1 | tee gpurun_out/r4/s1_sustained.txt
