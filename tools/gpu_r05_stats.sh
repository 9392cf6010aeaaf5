#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the bench command (5 + 2 steps) -> gpurun_out/r05_bench_kernel_stats.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R
python tools/prof_stats.py gpurun_out/prof/bench_results.db 60 > gpurun_out/r05_bench_kernel_stats.txt
rm -rf gpurun_out/prof
cat gpurun_out/r05_bench_kernel_stats.txt | cut -c1-140
