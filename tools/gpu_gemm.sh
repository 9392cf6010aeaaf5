#!/bin/bash
export TMPDIR=/tmp
PYTHONPATH=. python tools/dbg/gemm_mid.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_frontend_nn.py -q -x -m gpu -k "gemm or full_size or large_batch" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --ce --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done
