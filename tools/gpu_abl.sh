#!/bin/bash
# GPU box: kernel stats of `bench.py --den-only` for experiment builds (PK2_LIB selects libpk2hip_<tag>.so).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for tag in "$@"; do
  lib=$R/pykaldi2_amd/libpk2hip_$tag.so; [ "$tag" = base ] && lib=$R/pykaldi2_amd/libpk2hip.so
  PK2_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abl_$tag -o den -- python $R/bench.py --den-only > $R/gpurun_out/abl_$tag.log 2>&1
  echo "== $tag"; python $R/tools/prof_stats.py $R/gpurun_out/abl_$tag/den_results.db 4 | cut -c1-150; grep -o '"ms_per_launch": [0-9.]*' $R/gpurun_out/abl_$tag.log
  rm -rf $R/gpurun_out/abl_$tag
done
