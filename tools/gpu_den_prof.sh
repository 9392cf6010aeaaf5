#!/bin/bash
# GPU box: rocprofv3 kernel stats of `bench.py --den-only` for both den-graph topologies.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for topo in chain unique; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/den_trace_$topo -o den -- python $R/bench.py --den-only --den-topology $topo > $R/gpurun_out/den_trace_$topo.log 2>&1
  python $R/tools/prof_stats.py $R/gpurun_out/den_trace_$topo/den_results.db 14 > $R/gpurun_out/den_kernel_stats_$topo.txt
  grep -o '{"bound.*' $R/gpurun_out/den_trace_$topo.log >> $R/gpurun_out/den_kernel_stats_$topo.txt
  rm -rf $R/gpurun_out/den_trace_$topo
  cat $R/gpurun_out/den_kernel_stats_$topo.txt
done
