#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt gpurun_out/den_ab.txt
for f in 1 2 1 2; do echo "== PK2_DEN_PERSIST=$f" >> gpurun_out/den_ab.txt; PK2_DEN_PERSIST=$f timeout 300 python bench.py --den-only 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'], d['us_per_frame'], d['persist_form'])" >> gpurun_out/den_ab.txt; done
cat gpurun_out/den_ab.txt
timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/chain.log 2>&1; echo "chain exit $?" >> gpurun_out/summary.txt
tail -3 gpurun_out/chain.log | cut -c1-250
cat gpurun_out/summary.txt
