#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt gpurun_out/ride.txt
timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/chain.log 2>&1; echo "chain exit $?" >> gpurun_out/summary.txt
tail -3 gpurun_out/chain.log | cut -c1-250
for r in 0 1 0 1; do echo "== PK2_DEN_NUM_RIDE=$r" >> gpurun_out/ride.txt; PK2_DEN_NUM_RIDE=$r timeout 300 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['breakdown_ms'], d['parity']['ok'], d['parity']['objective_rel_err'], d['parity']['grad_max_abs_err'])" >> gpurun_out/ride.txt; done; cat gpurun_out/ride.txt
timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -3
cat gpurun_out/summary.txt
