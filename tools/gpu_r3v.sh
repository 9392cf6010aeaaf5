#!/bin/bash
export TMPDIR=/tmp
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_bgp.so timeout 300 python bench.py --ce --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^lstm_.wd_big" | sort | uniq -c | sort -rn | head -8
