#!/bin/bash
# the lattice-MMI child of the default bench line as the parent starts it, alone (twice, workspace cache on / off), then the whole
# default line REPS times: the child's step time, its host time per step and the hipMalloc calls inside its timed region
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
for c in 1 0; do
  PK2_LAT_WS_CACHE=$c timeout 600 python bench.py --se --steps 3 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('child alone, cache $c:', d['ms_per_step'], d['value'], d.get('step_host_ms'), d.get('device_allocs_in_timed_region'))"
done
for j in $(seq ${REPS:-3}); do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']['se']; print('default line', d['ms_per_step'], 'se', s['ms_per_step'], s.get('step_host_ms'), s.get('device_allocs_in_timed_region'), s.get('lattice_ms'))"
done
