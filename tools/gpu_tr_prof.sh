cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --transformer --steps 3 --warmup 2 --no-cpu-baseline > /tmp/prof_tr.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_stats.py /tmp/prof_tr/tr_results.db 26 | cut -c1-160; grep '"metric"' /tmp/prof_tr.log | cut -c1-400
