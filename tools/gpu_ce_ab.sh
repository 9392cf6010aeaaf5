#!/bin/bash
# CE step: kernel statistics of bench.py --ce (softmax_ce_kernel and what surrounds it), optionally by PK2_CE_GRID
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for grid in ${GRIDS:-2048}; do
  rm -rf /tmp/prof_ce
  PK2_CE_GRID=$grid timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ce -o ce -- python $R/bench.py --ce --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_ce.log 2>&1
  echo "== grid $grid"; python $R/tools/prof_stats.py /tmp/prof_ce/ce_results.db 16 | grep "softmax_ce\|scale_by\|elementwise\|total kernel" | cut -c1-150
  grep -o '"ms_per_step": [0-9.]*' /tmp/prof_ce.log | head -1
done
