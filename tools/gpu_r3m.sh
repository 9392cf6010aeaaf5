#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_bgp.so timeout 300 python bench.py --ce --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "big_persist" | sort -k1,1 | awk 'NR%8==1' | head -4 | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_frontend_nn.py -m gpu -q -x --tb=short -p no:cacheprovider -k "large_batch or full_size_ce" 2>&1 | tail -3
for f in 0 1; do PK2_LSTM_BIG_PERSIST=$f timeout 300 python bench.py --ce --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; done
