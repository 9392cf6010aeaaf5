#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt gpurun_out/den_ab.txt
for f in 1 2; do echo "== profile build, PK2_DEN_PERSIST=$f" >> gpurun_out/den_ab.txt; PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so PK2_DEN_PERSIST=$f timeout 300 python bench.py --den-only 2>/dev/null | grep "^den_persist" | tail -2 | cut -c1-700 >> gpurun_out/den_ab.txt; done
for f in 1 2 1 2; do echo "== PK2_DEN_PERSIST=$f" >> gpurun_out/den_ab.txt; PK2_DEN_PERSIST=$f timeout 300 python bench.py --den-only 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'], d['us_per_frame'], d['persist_form'])" >> gpurun_out/den_ab.txt; done
cat gpurun_out/den_ab.txt
timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/chain.log 2>&1; echo "chain exit $?" >> gpurun_out/summary.txt
tail -5 gpurun_out/chain.log | cut -c1-250
SWEEP_MODES="default frames" SWEEP_S="10000 30000 50000" SWEEP_A="500000 1000000 1500000 2000000" bash tools/gpu_den_sweep.sh
cat gpurun_out/summary.txt
