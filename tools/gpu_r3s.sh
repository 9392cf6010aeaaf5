#!/bin/bash
export TMPDIR=/tmp
for A in 1000000 1500000; do echo "== A=$A"; PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so timeout 300 python bench.py --den-only --den-states 30000 --den-arcs $A 2>/dev/null | grep "^den_persist2" | tail -2 | cut -c1-560; done
