#!/bin/bash
# GPU box: TransformerAM after a change -- its parity tests and the --transformer line with the weight gradients on the side
# stream (default) and on one stream (PK2_TR_SIDE_STREAM=0).
mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_transformer.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
for v in 1 0 1 0; do
PK2_TR_SIDE_STREAM=$v timeout 600 python bench.py --transformer --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side=$v', d['value'], d['ms_per_step'], d.get('breakdown_ms'), d['roofline_model']['frac'])"
done
} > gpurun_out/r05_tr.txt 2>&1
cat gpurun_out/r05_tr.txt | cut -c1-250
