#!/bin/bash
# idle time between the kernels of the LF-MMI step: kernel trace of 16 timed steps, gaps by (kernel before, kernel after)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; rm -rf /tmp/prof_g
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_g -o b -- python $R/bench.py --gpus 1 --steps 16 --warmup 3 --no-cpu-baseline --no-secondary > /tmp/prof_g.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_g.log | head -1
python $R/tools/gap_stats.py /tmp/prof_g/b_results.db 3 | cut -c1-140 | head -12; python $R/tools/step_sequence.py /tmp/prof_g/b_results.db 5 | cut -c1-150
