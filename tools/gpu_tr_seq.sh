#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tr -o tr -- python $R/bench.py --transformer --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_tr.log 2>&1
cd $R
python tools/prof_stats.py gpurun_out/prof_tr/tr_results.db 30 > gpurun_out/tr_kernel_stats.txt
python tools/step_sequence.py gpurun_out/prof_tr/tr_results.db > gpurun_out/tr_step_sequence.txt
rm -rf gpurun_out/prof_tr
cat gpurun_out/tr_kernel_stats.txt; tail -3 gpurun_out/tr_step_sequence.txt
