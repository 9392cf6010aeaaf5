#!/bin/bash
# GPU box: in-kernel phase timers and per-rank timeline of the persistent denominator (profile build libpk2hip_dpp.so =
# -DPK2_DP_PROFILE) on the fixed roofline workload.
mkdir -p gpurun_out; export TMPDIR=/tmp
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so timeout 300 python bench.py --den-only > /tmp/o.txt 2>&1
grep "^den_persist\|^timeline\|^  [ 0-9][0-9]:" /tmp/o.txt | tail -80 > gpurun_out/den_phases.txt
grep -o '"ms_per_launch": [0-9.]*' /tmp/o.txt >> gpurun_out/den_phases.txt
cat gpurun_out/den_phases.txt | cut -c1-420
