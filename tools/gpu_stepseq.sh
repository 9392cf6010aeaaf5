#!/bin/bash
# kernel sequence of one LF-MMI step (rocprofv3 --kernel-trace) + per-kernel statistics
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_seq -o seq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_seq.log 2>&1
cd $R
python tools/step_sequence.py gpurun_out/prof_seq/seq_results.db > gpurun_out/step_sequence.txt
python tools/prof_stats.py gpurun_out/prof_seq/seq_results.db 24 > gpurun_out/bench_kernel_stats.txt
rm -rf gpurun_out/prof_seq
cat gpurun_out/step_sequence.txt
