cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_se -o se -- python $GRAFT_REPO_ROOT/bench.py --se --steps 3 --warmup 2 --no-cpu-baseline > /tmp/prof_se.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_stats.py /tmp/prof_se/se_results.db 22 | cut -c1-160; grep '"metric"' /tmp/prof_se.log | cut -c1-400
