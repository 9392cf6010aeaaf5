#!/bin/bash
# denominator call alone (bench.py --den-only: 4 sequences, 1878 frames) under the default library and an experiment build, alternating
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-dtid}; REPS=${REPS:-3}
cd $R
for rep in $(seq $REPS); do
  for lib in libpk2hip.so libpk2hip_$TAG.so; do
    PK2_LIB=$R/pykaldi2_amd/$lib timeout 300 python bench.py --den-only 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_launch'], d['us_per_frame'])"
  done
done
