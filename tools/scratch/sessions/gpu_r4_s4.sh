#!/bin/bash
# GPU box, round 4 session 4: shared stage-0 ring in column blocks (420 / 600), cleaned pair kernel, f32 bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4/s4_pytest.txt
tools/gpu_var.sh r3 base r3 base 2>&1 | tee gpurun_out/r4/s4_ab.txt
for rep in 1 2; do
python bench.py --side 600 --batch 64 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('600 base %.0f img/s  cold %.0f  ' % (d['value'], d.get('cold_images_per_sec', 0)) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + ' head %.3f' % d['path']['head_ms'] + '  parity %s' % (d['parity'].get('ids_wrong'),))"
done 2>&1 | tee gpurun_out/r4/s4_600.txt
python bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --handles 1 --spinup-steps 0 2>&1 | tail -1 > gpurun_out/r4/s4_f32_bench.json
