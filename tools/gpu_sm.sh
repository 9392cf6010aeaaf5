#!/bin/bash
# GPU box: plain (XCD-local) exchange stores in the persistent denominator and the large-batch recurrences against the
# agent-scope build (libpk2hip_old.so = -DPK2_DP_STOREMODE=1 -DPK2_BIG_STOREMODE=1).
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for lib in _old "" _old ""; do
  echo "== lib$lib den-only"
  PK2_LIB=$PWD/pykaldi2_amd/libpk2hip$lib.so timeout 300 python bench.py --den-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('ms_per_launch','us_per_frame','persist_form') if k in d})"
done
for lib in _old "" _old ""; do
  echo "== lib$lib ce"
  PK2_LIB=$PWD/pykaldi2_amd/libpk2hip$lib.so timeout 300 python bench.py --ce --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'), d.get('parity'))"
done
timeout 1200 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_frontend_nn.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
} > gpurun_out/sm.txt 2>&1
cat gpurun_out/sm.txt | cut -c1-400
