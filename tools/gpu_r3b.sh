#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt
lscpu | grep -E "Model name|^CPU\(s\)|MHz" > gpurun_out/host.txt; nproc >> gpurun_out/host.txt
t() { n=$1; shift; timeout 900 python -m pytest "$@" -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/$n.log 2>&1; echo "$n exit $?" >> gpurun_out/summary.txt; tail -4 gpurun_out/$n.log | cut -c1-300; }
t tr_new tests/test_gpu_transformer.py -k "reference_golden"
t chain_new tests/test_gpu_chain.py -k "bench_size"
timeout 600 python tools/host_time.py > gpurun_out/host_time.txt 2>&1; head -45 gpurun_out/host_time.txt | cut -c1-200
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/summary.txt
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms'], d['parity'])"
cat gpurun_out/host.txt gpurun_out/summary.txt
