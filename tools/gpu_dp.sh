#!/bin/bash
# persistent denominator: parity + timing, in-kernel phase timers, kernel trace
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 200 python tools/dbg/den_persist_cmp.py ${DP_WHICH:-small big} 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-300
if [ -f pykaldi2_amd/libpk2hip_dpp.so ]; then
  PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so timeout 200 python tools/dbg/den_persist_cmp.py big 2>&1 | grep -A34 "den_persist\|timeline" | tail -72
fi
if [ -n "$DP_TRACE" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dpt -o dp --output-format csv -- env PYTHONPATH=$GRAFT_REPO_ROOT python $GRAFT_REPO_ROOT/tools/dbg/den_persist_cmp.py big > /tmp/dpt.log 2>&1
  f=$(find /tmp/dpt -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -14 "$f" | cut -c1-200 || { find /tmp/dpt | head; tail -5 /tmp/dpt.log; }
fi
