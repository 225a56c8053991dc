#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
echo "== product library"; timeout 900 python tools/scratch/dbg600.py
echo "== carried rounding for conv stages 1..6 only"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_upto6.so timeout 900 python tools/scratch/dbg600.py
timeout 900 python -m pytest tests/test_hip_fused.py -m gpu -q -k "cross_stage_fusion_is_bit_identical_to_stage_launches or one_launch_back_end_is_bit" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-600 | head -30
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="new (dither)"; line
V="new --no-dither"; line --no-dither
V="r5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line
done
} > gpurun_out/r6/s8.log 2>&1
tail -60 gpurun_out/r6/s8.log
