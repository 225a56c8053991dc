#!/usr/bin/env python3
"""Per-launch HBM traffic from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) vs the stage-boundary byte model.
usage: tools/hbm_traffic.py <fetch.csv> <write.csv> [batch side dtype] > profiles/<tag>_hbm_traffic.json
Counters are KiB per dispatch; gfx950: FETCH_SIZE under-reports 16 B/lane streaming reads by exactly 2x (MI355X_MICROARCH.md).
A cross-stage fused launch (stage23pc_kernel) is listed with the stages it computes: its algorithmic bytes are the
stage-boundary model's bytes of both stages (what two launches would move), its traffic is what it really moves."""
import csv, json, sys, collections, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from roomnet_amd.graph import build_graph
def per_kernel(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        agg.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return agg
fetch = per_kernel(sys.argv[1], 'FETCH_SIZE'); write = per_kernel(sys.argv[2], 'WRITE_SIZE')
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
side = int(sys.argv[4]) if len(sys.argv) > 4 else 224
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
elem = 2
g = build_graph(6, side)
def model_bytes(i):
    s = g.stages[i]
    alg = s.in_side * s.in_side * s.cin * (1 if i == 0 else elem) + s.out_side * s.out_side * s.cout * elem
    if s.residual: alg += s.skip_side * s.skip_side * s.cout * elem
    return alg * B
stage_kernels = [k for k in fetch if 'stage' in k or 'tail_kernel' in k or 'conv16' in k or 'backend_kernel' in k]
# the parity check in front of the timed passes runs 40 images once: below half a chip of images the back end runs as its three
# banded launches (stage6x / conv16p / tail), which then show up with ONE dispatch each: not part of the timed pass
most = max(len(fetch[k]) for k in stage_kernels)
stage_kernels = [k for k in stage_kernels if 2 * len(fetch[k]) >= most]
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 3 --warmup 1` (batch %d, "
               "%dx%d, %s). KiB per dispatch, mean over dispatches. gfx950 correction: FETCH_SIZE doubled (wide streaming reads are "
               "tallied at half size), WRITE_SIZE exact. traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; algorithmic_bytes = "
               "stage-boundary model (SURVEY.md 8d) x batch, summed over the stages a launch computes." % (B, side, side, dtype),
       "batch": B, "im_side": side, "dtype": dtype, "stages": []}
i = 0
for k in stage_kernels:
    if i >= len(g.stages): break
    if 'backend_kernel' in k:
        stages = list(range(i, len(g.stages)))      # stage 6 .. head in one launch (rn_backend.hip)
    elif 'stage23' in k or 'tail_kernel' in k or (i == 0 and 'stage0' not in k):
        stages = [i, i + 1]          # cross-stage fused launches: stages 2+3, the tail (8+9+head), stage 0 inside stage 1
    else:
        stages = [i]
    alg = sum(model_bytes(j) for j in stages)
    f = sum(fetch[k]) / len(fetch[k]); w = sum(write[k]) / len(write[k])
    t = int((2 * f + w) * 1024)
    out["stages"].append({"stage": stages[-1], "stages": stages, "kernel": (k[k.find('stage'):] if 'stage' in k else k[max(k.find('conv16'), k.find('backend_kernel'), 0):])[:60], "fetch_size_kib": f, "write_size_kib": w,
                          "traffic_bytes": t, "algorithmic_bytes": alg, "traffic_over_algorithmic": round(t / alg, 3)})
    i = stages[-1] + 1
print(json.dumps(out, indent=1))
