#!/bin/bash
# where the device idles inside the headline step: kernel trace of the default bench, gaps between consecutive kernels
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; rm -rf /tmp/prof_b
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /tmp/prof_b.log 2>&1
python $R/tools/gap_stats.py /tmp/prof_b/bench_results.db 3 | head -24
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_b.log | head -2
