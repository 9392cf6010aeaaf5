#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R
python tools/prof_stats.py gpurun_out/prof/bench_results.db 45 > gpurun_out/bench_kernel_stats.txt
cat gpurun_out/bench_kernel_stats.txt | cut -c1-160
