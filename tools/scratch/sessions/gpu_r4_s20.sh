#!/bin/bash
# tail constants pre-staged in LDS: parity + back-end / tail timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_fused.py tests/test_roomnet_api.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tee gpurun_out/r4/s20_pytest.txt
STEPS=400 tools/gpu_var.sh r3 base base 2>&1 | tee gpurun_out/r4/s20_ab.txt
python tools/gpu_batch_sweep_one.py 2>/dev/null || true
for b in 1 8 64; do python bench.py --batch $b --steps 2000 --warmup 50 --no-cpu-baseline --no-parity-check --no-cold-pass 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('batch', d['config']['global_batch'], 'ms/step %.4f' % d['ms_per_step'])"; done | tee -a gpurun_out/r4/s20_ab.txt
