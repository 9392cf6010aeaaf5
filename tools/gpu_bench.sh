#!/bin/bash
# GPU box: bench line + rocprofv3 kernel stats of the same command (short run).
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-10}
timeout 900 python bench.py --gpus 1 --steps $STEPS --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/summary.txt
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
echo "rocprof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.txt
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
# keep the merged output small: drop the raw trace, keep stats
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
cat gpurun_out/summary.txt
