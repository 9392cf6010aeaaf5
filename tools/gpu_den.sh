#!/bin/bash
# GPU box: chain parity tests + denominator timings on both graph topologies / both kernel families.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_chain.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/test_gpu_chain.log 2>&1
echo "chain tests exit $?" >> gpurun_out/summary.txt
tail -15 gpurun_out/test_gpu_chain.log
for topo in chain unique; do
  for mode in "" general; do
    echo "== topology $topo mode ${mode:-default}" >> gpurun_out/den_only.txt
    PK2_DEN_MODE=$mode timeout 300 python bench.py --den-only --den-topology $topo 2>/dev/null | tail -1 >> gpurun_out/den_only.txt
  done
done
cat gpurun_out/den_only.txt
cat gpurun_out/summary.txt
