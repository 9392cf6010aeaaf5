#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for lib in pykaldi2_amd/libpk2hip.so pykaldi2_amd/libpk2hip_v*.so; do
  echo "== $lib"
  PK2_LIB=$PWD/$lib timeout 120 python tools/dbg/den_persist_cmp.py big 2>&1 | grep "big" | tail -3
done
