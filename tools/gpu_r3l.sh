#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ce -o ce -- python $R/bench.py --ce --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_ce.log 2>&1
cd $R
python tools/prof_stats.py gpurun_out/prof_ce/ce_results.db 14 > gpurun_out/ce_kernel_stats.txt
rm -rf gpurun_out/prof_ce
cat gpurun_out/ce_kernel_stats.txt | cut -c1-150
