#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 1200 python -m pytest tests/test_gpu_frontend_nn.py tests/test_gpu_cli.py -m gpu -q -x --tb=short -p no:cacheprovider -k "not eight_ranks and not 50_steps" 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'), d['persistent_health'])"
done
} > gpurun_out/quick.txt 2>&1
cat gpurun_out/quick.txt | cut -c1-300
