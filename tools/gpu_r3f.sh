#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for f in 1 2; do echo "== profile build, PK2_DEN_PERSIST=$f"; PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so PK2_DEN_PERSIST=$f timeout 300 python bench.py --den-only 2>/dev/null | grep -v "^{" | tail -70 | cut -c1-400; done > gpurun_out/den_tl.txt
cat gpurun_out/den_tl.txt
