#!/bin/bash
# GPU box: the secondary workloads' bench lines (configs[1], [3], [4]).
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for w in se ce transformer; do
  timeout 600 python bench.py --$w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d.get('breakdown_ms'))"
done
} > gpurun_out/secondary.txt 2>&1
cat gpurun_out/secondary.txt | cut -c1-400
