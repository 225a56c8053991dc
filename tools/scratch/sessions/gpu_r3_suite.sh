#!/bin/bash
# GPU box, round 3: whole GPU test suite, parity report, directory-driver and group benchmarks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3/pytest_all.log
cp gpurun_out/parity_report.json gpurun_out/r3/parity.json 2>/dev/null
timeout 300 python tools/bench_images.py --dir 1080 1920 256 2>&1 | tail -1 | tee gpurun_out/r3/bench_images_dir.txt
timeout 300 python tools/bench_images.py --dir 480 640 512 2>&1 | tail -1 | tee -a gpurun_out/r3/bench_images_dir.txt
timeout 300 python tools/bench_images.py 1080 1920 256 2>&1 | tail -1 | tee -a gpurun_out/r3/bench_images_dir.txt
timeout 300 python tools/group_bench.py --gpus 1 2>/dev/null | tail -1 | tee gpurun_out/r3/group_bench_1.json | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/r3/bench_quick.json | cut -c1-300
