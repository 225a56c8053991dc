#!/bin/bash
# GPU box, round 4 session 13: column-blocked back end (stages 6 + 7 at 420 / 600): tests + 600 bench + 224 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_fused.py tests/test_roomnet_api.py -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r4/s13_pytest.txt
for rep in 1 2; do
python bench.py --side 600 --batch 64 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('600 base %.0f img/s  cold %.0f  ' % (d['value'], d.get('cold_images_per_sec', 0)) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + ' head %.3f' % d['path']['head_ms'] + '  parity %s' % (d['parity'].get('ids_wrong'),), d['path']['launch_groups'])"
done 2>&1 | tee gpurun_out/r4/s13_600.txt
tools/gpu_var.sh r3 base 2>&1 | tee gpurun_out/r4/s13_ab.txt
