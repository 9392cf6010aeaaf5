#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt
timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/chain.log 2>&1; echo "chain exit $?" >> gpurun_out/summary.txt
tail -4 gpurun_out/chain.log | cut -c1-250
SWEEP_MODES="default" SWEEP_S="10000 30000 50000" SWEEP_A="1000000 1500000 2000000" bash tools/gpu_den_sweep.sh
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/summary.txt
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms'], d['roofline']['ms_per_launch'], d['roofline']['kernel'][:30], d['parity']['ok'])"
for f in tests/test_gpu_comm.py tests/test_gpu_frontend_nn.py tests/test_gpu_cli.py tests/test_gpu_transformer.py tests/test_gpu_lattice.py; do n=$(basename $f .py); timeout 900 python -m pytest $f -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/$n.log 2>&1; echo "$n exit $?" >> gpurun_out/summary.txt; tail -2 gpurun_out/$n.log | cut -c1-200; done
cat gpurun_out/summary.txt
