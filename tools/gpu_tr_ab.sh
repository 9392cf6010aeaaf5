#!/bin/bash
# TransformerAM step under the default library and an experiment build (TAG), same box: kernel statistics of the attention kernels
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-dq1}; cd /tmp
for rep in 1 2; do
  for lib in libpk2hip.so libpk2hip_$TAG.so; do
    rm -rf /tmp/prof_tr
    PK2_LIB=$R/pykaldi2_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $R/bench.py --transformer --steps 8 --warmup 3 --no-cpu-baseline > /tmp/prof_tr.log 2>&1
    echo "== $lib"; python $R/tools/prof_stats.py /tmp/prof_tr/tr_results.db 14 | grep "attn_" | cut -c1-150
    grep -o '"ms_per_step": [0-9.]*' /tmp/prof_tr.log | head -1
  done
done
