#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt
t() { n=$1; shift; timeout 1800 python -m pytest "$@" -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/$n.log 2>&1; echo "$n exit $?" >> gpurun_out/summary.txt; tail -12 gpurun_out/$n.log | cut -c1-300; }
t recipe tests/test_gpu_recipe.py
t cli_new tests/test_gpu_cli.py -k "overlap_schedule_50 or eight_ranks"
t chain_new tests/test_gpu_chain.py -k "beyond_the_register"
SWEEP_MODES="default" SWEEP_S="50000" SWEEP_A="500000 1000000" bash tools/gpu_den_sweep.sh
cat gpurun_out/summary.txt
