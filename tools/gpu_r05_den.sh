#!/bin/bash
# GPU box: A/B of the denominator kernel variants (libpk2hip_<tag>.so next to the default build) on the fixed --den-only
# workload, then the chain parity tests and a bench line.  VARIANTS="noasm ..." names the tags.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for v in default ${VARIANTS}; do
  lib=$PWD/pykaldi2_amd/libpk2hip.so; [ $v != default ] && lib=$PWD/pykaldi2_amd/libpk2hip_$v.so
  for i in 1 2; do
    PK2_LIB=$lib timeout 300 python bench.py --den-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_launch'], d['us_per_frame'], d['persist_form'])"
  done
done
timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'), d['persistent_health'])"
done
} > gpurun_out/r05_den_ab.txt 2>&1
cat gpurun_out/r05_den_ab.txt | cut -c1-300
