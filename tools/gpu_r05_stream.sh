#!/bin/bash
# GPU box: phase timers of the streaming variant of the persistent denominator (profile build libpk2hip_dpp.so,
# -DPK2_DP_PROFILE) on S = 30 k graphs of 1.0 / 1.5 / 2.0 M arcs, with the layouts the builder chose.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for A in ${STREAM_A:-1000000 1500000 2000000}; do
  echo "## S=30000 A=$A"
  PK2_DP2_DEBUG=1 PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so timeout 300 python bench.py --den-only --den-states 30000 --den-arcs $A > /tmp/o.txt 2>&1
  grep "^persist2:" /tmp/o.txt | sort | uniq -c; grep "^den_persist2" /tmp/o.txt | tail -2
done
} > gpurun_out/r05_den_stream_phases.txt 2>&1
cat gpurun_out/r05_den_stream_phases.txt
