#!/bin/bash
# large-batch persistent recurrence: parity tests + the CE bench line + phase timers
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_frontend_nn.py -q -x -m gpu -k "large_batch or full_size" 2>&1 | tail -5
for i in 1 2; do timeout 300 python bench.py --ce --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('breakdown_ms'))"; done
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_bgp.so timeout 300 python bench.py --ce --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^lstm_.wd_big" | sort | uniq -c | sort -rn | sed -n '1,3p;40,42p' | cut -c1-330
