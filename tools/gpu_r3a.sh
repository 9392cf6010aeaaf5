#!/bin/bash
# GPU box, round 3 first call: the new bench-size parity tests, the touched LSTM paths, bench lines with `parity`.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.txt
t() { n=$1; shift; timeout 900 python -m pytest "$@" -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/$n.log 2>&1; echo "$n exit $?" >> gpurun_out/summary.txt; tail -8 gpurun_out/$n.log | cut -c1-300; }
t chain_new tests/test_gpu_chain.py -k "bench_size"
t nn_new tests/test_gpu_frontend_nn.py -k "full_size_ce or persistent_recurrence or blstm_3x512"
t tr_new tests/test_gpu_transformer.py -k "reference_golden"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/summary.txt
cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
timeout 600 python bench.py --ce --steps 10 --warmup 3 > gpurun_out/bench_ce.json 2> gpurun_out/bench_ce.err; echo "bench ce exit $?" >> gpurun_out/summary.txt
cat gpurun_out/bench_ce.json; tail -3 gpurun_out/bench_ce.err
cat gpurun_out/summary.txt
