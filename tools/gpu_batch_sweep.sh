#!/bin/bash
# GPU box: throughput / latency over the batch size for the three handle types (bench.py, inputs resident in HBM) -> gpurun_out/r6/batch_sweep.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
echo "# bench.py --batch B --dtype D --steps K --warmup 5 (one MI355X, 224 x 224, inputs resident in HBM): ms per forward pass = latency of one call"
echo "# dtype batch   img/s     ms/pass   launch groups"
for dt in bf16 f16 f32; do for b in 1 2 8 32 128 256; do
  k=200; [ $dt = f32 ] && k=40
  python bench.py --batch $b --dtype $dt --steps $k --warmup 5 --no-cpu-baseline --no-other-configs --no-unfolded-arm --profile-steps 1 --event-steps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-5s %4d %9.0f   %8.4f   %s' % ('$dt', $b, d['value'], d['ms_per_step'], d['path']['launch_groups']))"
done; done
} 2>&1 | tee gpurun_out/r6/batch_sweep.txt
