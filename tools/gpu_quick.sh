#!/bin/bash
# GPU box: chain parity tests, a few sweep points, two bench lines (quick regression after a change of the LF-MMI path).
mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
SWEEP_S="${SWEEP_S:-30000 40000 50000}" SWEEP_A="${SWEEP_A:-1000000 1500000}" SWEEP_MODES=default bash tools/gpu_den_sweep.sh 2>/dev/null
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'), d['persistent_health'])"
done
} > gpurun_out/quick.txt 2>&1
cat gpurun_out/quick.txt | cut -c1-300
