#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt
timeout 900 python -m pytest tests/test_gpu_frontend_nn.py -m gpu -q -x --tb=short -p no:cacheprovider -k "large_batch or full_size_ce" > gpurun_out/nn.log 2>&1; echo "nn exit $?" >> gpurun_out/summary.txt; tail -12 gpurun_out/nn.log | cut -c1-300
for f in 0 1; do echo "== PK2_LSTM_BIG_PERSIST=$f" >> gpurun_out/ce_ab.txt; PK2_LSTM_BIG_PERSIST=$f timeout 300 python bench.py --ce --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss'])" >> gpurun_out/ce_ab.txt; done; cat gpurun_out/ce_ab.txt
cat gpurun_out/summary.txt
