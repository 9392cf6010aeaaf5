#!/bin/bash
# bf16x3 GEMM: parity at one bound for both arithmetic paths, error table, speed A/B of build variants
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_frontend_nn.py -q -x -m gpu -k "gemm" 2>&1 | tail -5
PYTHONPATH=. timeout 600 python tools/dbg/gemm_arith.py err 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x3_err.txt
for tag in "" _d4 _d4ns _nopin; do
  echo "== variant libpk2hip$tag.so"
  PK2_LIB=$PWD/pykaldi2_amd/libpk2hip$tag.so PYTHONPATH=. timeout 600 python tools/dbg/gemm_arith.py speed 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x3_speed$tag.txt
done
