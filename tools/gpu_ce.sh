#!/bin/bash
# large-batch persistent recurrence: parity tests, CE bench A/B of the two backward forms, phase timers (profile build)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_frontend_nn.py -q -x -m gpu -k "large_batch or full_size" 2>&1 | tail -3
for v in 0 1 0 1; do
  echo "== PK2_LSTM_BIG_BWD=$v"
  PK2_LSTM_BIG_BWD=$v timeout 300 python bench.py --ce --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
[ -f pykaldi2_amd/libpk2hip_bgp.so ] && PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_bgp.so timeout 300 python bench.py --ce --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^lstm_bwd_big" | sort | uniq -c | sort -rn | sed -n '1,2p;20,21p' | cut -c1-360
