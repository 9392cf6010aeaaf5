#!/bin/bash
# lat_frames_finish under the default library and experiment builds (TAGS), same box, REPS times each: per-kernel averages of the lattice-MMI step
#   PK2_BUILD_TAG=<tag> PK2_EXTRA_FLAGS="-DPK2_FIN_EPS=4 -DPK2_FIN_EMIT=16" python -m pykaldi2_amd.build;  TAGS="fin416 fin614" REPS=2 bash tools/gpu_fin_ab.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; REPS=${REPS:-2}
cd /tmp
for rep in $(seq $REPS); do
  for tag in "" $TAGS; do
    lib=libpk2hip${tag:+_$tag}.so
    rm -rf /tmp/prof_se
    PK2_LIB=$R/pykaldi2_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_se -o se -- python $R/bench.py --se --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_se.log 2>&1
    echo "== $lib"; python $R/tools/prof_stats.py /tmp/prof_se/se_results.db 8 | grep "lat_frames_finish\|lat_frames_persist\|total kernel"  | cut -c1-150
  done
done
