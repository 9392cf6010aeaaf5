#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for tag in "" $X3_TAGS; do
  echo "== variant libpk2hip$tag.so"
  for sh in "0 1 20480 4096 1024" "0 0 20480 1024 4096" "1 0 4096 1024 20480" "0 1 2276 4096 1024"; do
    PK2_LIB=$PWD/pykaldi2_amd/libpk2hip$tag.so PYTHONPATH=. timeout 300 python tools/dbg/gemm_one.py $sh 2>&1 | grep -v amdgpu.ids
  done
done
