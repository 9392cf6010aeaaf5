#!/bin/bash
# lattice forward-backward: parity tests, then kernel statistics of bench.py --se with the linear-domain recursion (default) and the
# log-add kernel (PK2_FB_LINEAR=0), same box; phase timers of the log-add kernel if a -DPK2_FB_PROFILE build (TAG fbp) is there
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
[ -z "$SKIP_TESTS" ] && timeout 1500 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_recipe.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3
[ -f pykaldi2_amd/libpk2hip_fbp.so ] && PK2_LIB=$R/pykaldi2_amd/libpk2hip_fbp.so timeout 300 python bench.py --se --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep "^lat_fb" | sort | uniq | head -3 | cut -c1-330
cd /tmp
for rep in 1 2; do
  for lin in 1 0; do
    rm -rf /tmp/prof_se
    PK2_FB_LINEAR=$lin timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_se -o se -- python $R/bench.py --se --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_se.log 2>&1
    echo "== PK2_FB_LINEAR=$lin"; python $R/tools/prof_stats.py /tmp/prof_se/se_results.db 12 | grep "lat_fb\|lat_frames_finish" | cut -c1-150
    grep -o '"ms_per_step": [0-9.]*' /tmp/prof_se.log | head -1
  done
done
