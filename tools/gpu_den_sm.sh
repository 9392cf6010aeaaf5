#!/bin/bash
# GPU box: the persistent denominator with plain (XCD-local) exchange stores (-DPK2_DP_STOREMODE=0) against the default.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for lib in "" _dps0; do
  echo "== lib$lib den-only"
  PK2_LIB=$PWD/pykaldi2_amd/libpk2hip$lib.so timeout 300 python bench.py --den-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('ms_per_launch','us_per_frame','kernel','persist_form') if k in d})"
done
for i in 1 2 3; do
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dps0.so timeout 1200 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
done
} > gpurun_out/den_sm.txt 2>&1
cat gpurun_out/den_sm.txt | cut -c1-300
