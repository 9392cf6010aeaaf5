#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/summary.txt
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_recipe.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/recipe.log 2>&1; echo "recipe exit $?" >> gpurun_out/summary.txt; tail -3 gpurun_out/recipe.log
cd /tmp
for ss in 0 1; do
  PK2_SIDE_STREAM=$ss timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/side$ss -o b -- python $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/side$ss.log 2>&1
  echo "== PK2_SIDE_STREAM=$ss" >> $R/gpurun_out/side_overlap.txt
  grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/side$ss.log >> $R/gpurun_out/side_overlap.txt
  python $R/tools/overlap_stats.py $R/gpurun_out/side$ss/b_results.db 12 >> $R/gpurun_out/side_overlap.txt
  rm -rf $R/gpurun_out/side$ss
done
cd $R; cat gpurun_out/side_overlap.txt; cat gpurun_out/summary.txt
