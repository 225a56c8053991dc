#!/bin/bash
# GPU box: the measurement ladder of VERDICT r3 item 4 on ONE box: 20 / 200 / 6000 timed steps x spin-up (240 / 0) x handles (1 / 2)
# x dtype (bf16 / f16), every run a fresh process (so every run starts from an idle chip).  -> gpurun_out/r4/ladder.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
{
echo "# tools/gpu_ladder.sh on one MI355X box: bench.py --steps K --warmup 5 --spinup-steps S --handles H --dtype D (batch 256, 224 x 224)"
echo "# value = K timed steps after the spin-up and warm-up; cold = the same W + K steps timed BEFORE the spin-up on one handle"
echo "# dtype handles spinup steps   value img/s   ms/step   cold img/s   path/HBM-roofline(value)"
for dt in bf16 f16; do for h in 1 2; do for sp in 240 0; do for k in 20 200 6000; do
  python bench.py --steps $k --warmup 5 --spinup-steps $sp --handles $h --dtype $dt --no-cpu-baseline --profile-steps 1 --event-steps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-5s %d %4d %5d   %9.0f   %.4f   %9.0f   %.4f' % ('$dt', $h, $sp, $k, d['value'], d['ms_per_step'], d.get('cold_images_per_sec', 0), d['path']['hbm_frac']))"
done; done; done; done
} 2>&1 | tee gpurun_out/r4/ladder.txt
