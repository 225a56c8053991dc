cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_hip_fused.py -x -q -m gpu -k "cross_stage or stage_outputs or logits or band_and_batch" > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -30 gpurun_out/r2a/pytest.log
for arm in "" "--stage-launches"; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline $arm 2>gpurun_out/r2a/bench_err$arm.log | tail -1 > gpurun_out/r2a/bench$arm.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r2a/bench$arm.json"))
print("$arm", "%.0f img/s" % d["value"], " ".join("%.3f"%x for x in d["path"]["stage_ms"]), "head %.3f" % d["path"]["head_ms"])
PY
done
