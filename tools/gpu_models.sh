#!/bin/bash
# GPU box: model parity tests (front end, BLSTM, GEMM, TransformerAM), two headline bench lines, the secondary bench lines.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_frontend_nn.py tests/test_gpu_transformer.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'))"
done
bash tools/gpu_secondary.sh
