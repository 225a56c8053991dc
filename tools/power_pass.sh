#!/bin/bash
# GPU box: socket power and engine clock next to the RoomNet forward loop (batch 256, 224x224), and the energy per image they
# imply: tools/power_pass.sh > gpurun_out/r3/power.txt   (rocm-smi sampled once per second; the bench line's img/s)
cd $GRAFT_REPO_ROOT
smi() { rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Package Power" | sed -e 's/.*sclk clock level: [0-9S]*: //' -e 's/.*Power (W): /W /' | tr '\n' ' '; echo; }
for arm in "--dtype bf16" "--dtype f16" "--dtype bf16 --no-dither" "--dtype f32 --steps 600"; do
  echo "--- bench.py --steps 6000 $arm"
  T=$(mktemp)
  (while true; do echo "   [smi] $(smi)"; sleep 1; done) > $T &
  SP=$!
  L=$(python bench.py --steps 6000 --warmup 20 --no-cpu-baseline --no-parity-check --no-other-configs --no-unfolded-arm $arm 2>/dev/null | tail -1)
  kill $SP; wait $SP 2>/dev/null
  cat $T
  python3 - "$T" <<PY
import json, re, sys
d = json.loads('''$L''')
w = [float(m.group(1)) for m in re.finditer(r"W ([0-9.]+)", open(sys.argv[1]).read())]
w = [x for x in w if x > 0.8 * max(w)]          # samples taken under load
print("%.0f img/s, %.3f ms per step; mean socket power under load %.0f W (%d samples) -> %.2f mJ per image"
      % (d["value"], d["ms_per_step"], sum(w) / len(w), len(w), 1e3 * (sum(w) / len(w)) / d["value"]))
PY
  rm -f $T
done
