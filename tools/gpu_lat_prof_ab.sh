#!/bin/bash
# kernel statistics of the lattice-MMI step under the default library and an experiment build (TAG), same box, REPS times each
#   PK2_BUILD_TAG=<tag> [PK2_EXTRA_FLAGS=...] python -m pykaldi2_amd.build;  TAG=<tag> REPS=2 bash tools/gpu_lat_prof_ab.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-nomerge}; REPS=${REPS:-1}
cd /tmp
for rep in $(seq $REPS); do
  for lib in libpk2hip.so libpk2hip_$TAG.so; do
    rm -rf /tmp/prof_se
    PK2_LIB=$R/pykaldi2_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_se -o se -- python $R/bench.py --se --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_se.log 2>&1
    echo "== $lib"; python $R/tools/prof_stats.py /tmp/prof_se/se_results.db 6 | cut -c1-150
  done
done
