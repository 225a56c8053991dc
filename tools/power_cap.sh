#!/bin/bash
# GPU box: sustained MFMA / HBM rates at the power cap, and clock + power of the RoomNet forward pass (NOTES.md, rounds 1-2 section 5a)
cd $GRAFT_REPO_ROOT
smi() { rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Package Power" | sed -e 's/.*sclk clock level: [0-9S]*: //' -e 's/.*Power (W): /W /' | tr '\n' ' '; echo; }
(while true; do echo "   [smi] $(smi)"; sleep 1; done) &
SP=$!
./tools/ubench/power_cap ${1:-4}
echo "--- RoomNet forward pass, batch 256, 224x224, bf16 then f16 (bench.py loop)"
python bench.py --steps 6000 --warmup 20 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | cut -c1-160
python bench.py --steps 6000 --warmup 20 --no-cpu-baseline --no-parity-check --dtype f16 2>/dev/null | tail -1 | cut -c1-160
kill $SP
