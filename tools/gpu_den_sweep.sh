#!/bin/bash
# GPU box: denominator call time over graph sizes: S x A sweep, the default kernel choice, one job.
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/den_sweep.txt; : > $out
for S in ${SWEEP_S:-10000 30000 50000}; do
  for A in ${SWEEP_A:-500000 1000000 1500000 2000000}; do
    for mode in ${SWEEP_MODES:-default frames}; do
      env=""; [ "$mode" = frames ] && env="PK2_DEN_PERSIST=0"; [ "$mode" = v1 ] && env="PK2_DEN_PERSIST=1"; [ "$mode" = v2 ] && env="PK2_DEN_PERSIST=2"
      line=$(env $env timeout 300 python bench.py --den-only --den-states $S --den-arcs $A 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ms %.3f us/frame %.2f form %s' % (d['ms_per_launch'], d['us_per_frame'], d.get('persist_form')))
except Exception as e: print('failed', e)")
      echo "S=$S A=$A mode=$mode $line" | tee -a $out
    done
  done
done
