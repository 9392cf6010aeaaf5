#!/bin/bash
# persistent lattice decoder: kernel time of lat_frames_persist by team size (workgroups per utterance), same box
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for team in 16 8 32 16; do
  rm -rf /tmp/prof_se
  PK2_LAT_TEAM=$team timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_se -o se -- python $R/bench.py --se --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_se.log 2>&1
  echo "== team $team"; python $R/tools/prof_stats.py /tmp/prof_se/se_results.db 3 | grep lat_frames_persist | cut -c1-150
done
