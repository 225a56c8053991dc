#!/bin/bash
# GPU box: the narrow pair's consumer with five two-tap chunks (this tree) against six (tools/ab/libroomnet_hip_six.so), same session
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_fused.py -x -q -m gpu 2>&1 | tail -5
line() { python bench.py --steps ${STEPS:-300} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="five chunks"; line "$@"
V="six chunks"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_six.so line "$@"
done
V="five, 600 f16"; line --side 600 --batch 64 --dtype f16
V="six, 600 f16"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_six.so line --side 600 --batch 64 --dtype f16
for t in 16 32; do timeout 200 python tools/bench_images.py --dir --threads=$t 1080 1920 384 2>&1 | tail -1 | cut -c1-260; done
timeout 200 python tools/bench_images.py --dir --threads=16 480 640 1024 2>&1 | tail -1 | cut -c1-260
