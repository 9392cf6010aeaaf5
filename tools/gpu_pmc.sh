#!/bin/bash
# GPU box: HBM traffic counters for the denominator forward-backward (separate --pmc passes, no tracing domains
# other than kernel-trace), plus a kernel-trace/stats pass of the same command.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o den -- python $R/bench.py --den-only > $R/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c exit $?" >> $R/gpurun_out/summary.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/den_trace -o den -- python $R/bench.py --den-only > $R/gpurun_out/den_trace.log 2>&1
echo "den trace exit $?" >> $R/gpurun_out/summary.txt
cd $R; ls -R gpurun_out | head -30; cat gpurun_out/summary.txt
