#!/bin/bash
# why is fp16 storage 5 % faster than bf16?  x1 = bf16 build whose conv MFMAs of stages 2-5 are the f16 instruction (same bits in the
# operands, wrong numbers: timing only); x2 = bf16 build whose stages 2-5 pack their outputs with v_cvt_pk_f16_f32 (wrong numbers)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  STEPS=600 tools/gpu_var.sh base x1 x2
  python bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-parity-check --dtype f16 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-12s %.0f img/s  ' % ('f16 mode', d['value']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']))"
done
