#!/bin/bash
# GPU box: the whole -m gpu suite (what the driver runs at round end), smoke, and the default bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/gpu_suite.log 2>&1
echo "gpu suite exit $? in $(( $(date +%s) - t0 )) s"; tail -6 gpurun_out/gpu_suite.log | cut -c1-300
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cat gpurun_out/bench.json | cut -c1-1500
