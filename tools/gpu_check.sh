#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests + smoke; logs into gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
for f in ${TESTS:-tests/test_gpu_chain.py tests/test_gpu_comm.py tests/test_gpu_frontend_nn.py tests/test_gpu_cli.py tests/test_gpu_transformer.py tests/test_gpu_lattice.py}; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/$n.log 2>&1
  echo "$n exit $?" >> gpurun_out/summary.txt
  tail -25 gpurun_out/$n.log | cut -c1-300
done
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/summary.txt
tail -5 gpurun_out/smoke.log
cat gpurun_out/summary.txt
