#!/usr/bin/env python3
"""GPU box: put the parity evidence on file.  Runs the GPU parity tests that record their numbers (tests/conftest.py
`parity_record`: per-stage relative errors of the 16-bit path at 224 and 600, max |dlogit| against the fp64 goldens,
class-id disagreements of 4096 random images against the float32 HIP path with their margins) and writes them as JSON.

    python tools/parity_report.py [OUT.json]        (default gpurun_out/r3/parity.json; copy to profiles/rN_parity.json)

The comparison itself lives in tests/ -- this script only selects the tests and names the output file (the oracle is
test infrastructure: nothing outside tests/ imports it)."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r3", "parity.json"))
os.makedirs(os.path.dirname(out), exist_ok=True)
select = ("test_stage_outputs_vs_oracle or test_logits_probs_ids_vs_golden or test_random_4096_images_id_agreement or test_randomized_batch_256 "
          "or test_600_variant_vs_golden")
rc = subprocess.call([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_fused.py"), "-q", "-m", "gpu",
                      "-k", select], env=dict(os.environ, RN_PARITY_REPORT=out), cwd=root)
print("parity report: %s (pytest rc %d)" % (out, rc))
sys.exit(rc)
