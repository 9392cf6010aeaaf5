#!/bin/bash
# bench.py --se under different step / warm-up counts and both forward-backward kernels: what the first visit of a minibatch shape costs
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "1 3 1" "1 3 3" "0 3 1" "1 6 3"; do set -- $cfg
  PK2_FB_LINEAR=$1 timeout 600 python bench.py --se --steps $2 --warmup $3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('linear=$1 steps=$2 warmup=$3', d['ms_per_step'], d['value'], d.get('lattice_ms'))"
done
