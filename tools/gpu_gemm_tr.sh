#!/bin/bash
# GPU box: the f32 GEMM parity test, then its time on the TransformerAM's shapes (R = 2356 frames, dim 512, FFN 2048) and the built-in list.
mkdir -p gpurun_out; export TMPDIR=/tmp
SH="0,1,2356,512,512;0,1,2356,1536,512;0,1,2356,2048,512;0,1,2356,512,2048;0,0,2356,512,512;0,0,2356,512,1536;0,0,2356,2048,512;0,0,2356,512,2048;1,0,512,512,2356;1,0,1536,512,2356;1,0,2048,512,2356;1,0,512,2048,2356;0,1,2356,6048,512;0,0,2356,512,6048;1,0,6048,512,2356"
{
timeout 600 python -m pytest tests/test_gpu_frontend_nn.py -q -x -m gpu -k "gemm" -p no:cacheprovider 2>&1 | tail -2
for mode in default $GEMM_EXTRA; do
  echo "== $mode"
  e=""; [ "$mode" != default ] && e="$mode"
  env $e timeout 300 python bench.py --gemm-only --gemm-shapes "$SH" 2>/dev/null | grep "^gemm"
  env $e timeout 300 python bench.py --gemm-only 2>/dev/null | grep "^gemm"
done
} > gpurun_out/gemm_tr.txt 2>&1
cat gpurun_out/gemm_tr.txt
