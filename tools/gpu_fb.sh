#!/bin/bash
# lattice forward-backward: parity tests, phase timers of the profile build (-DPK2_FB_PROFILE, TAG fbp), kernel statistics of bench.py --se
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
[ -z "$SKIP_TESTS" ] && timeout 1500 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_recipe.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3
[ -f pykaldi2_amd/libpk2hip_fbp.so ] && PK2_LIB=$R/pykaldi2_amd/libpk2hip_fbp.so timeout 300 python bench.py --se --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep "^lat_fb" | head -2 | cut -c1-330
cd /tmp
for rep in 1 2; do
  rm -rf /tmp/prof_se
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_se -o se -- python $R/bench.py --se --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_se.log 2>&1
  python $R/tools/prof_stats.py /tmp/prof_se/se_results.db 6 | grep "lat_fb\|lat_frames_finish" | cut -c1-150
  grep -o '"ms_per_step": [0-9.]*' /tmp/prof_se.log | head -1
done
