#!/bin/bash
# denominator, non-temporal history stores / x loads apart (VERDICT r5 #2c): us per frame (two rounds) and WRITE / FETCH bytes per call
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out
for round in 1 2; do for tag in "" _ntl _nts _nt0; do
  echo "round $round lib$tag: $(PK2_LIB=$R/pykaldi2_amd/libpk2hip$tag.so python $R/bench.py --den-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'], d['us_per_frame'])")"
done; done
cd /tmp
for tag in "" _ntl _nts _nt0; do for c in FETCH_SIZE WRITE_SIZE; do
  PK2_LIB=$R/pykaldi2_amd/libpk2hip$tag.so timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/nt_$c$tag -o den -- python $R/bench.py --den-only > /dev/null 2>&1
  python - "$R/gpurun_out/nt_$c$tag/den_results.db" $c "lib$tag" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
s = cur.execute("select sum(counter_value) from pmc_events where counter_name = ?", (sys.argv[2],)).fetchone()[0]
n = cur.execute("select count(*) from pmc_events where counter_name = ? and name like '%den_tail1%'", (sys.argv[2],)).fetchone()[0]
print(sys.argv[3], sys.argv[2], "GB per call: %.3f" % (s * 1024 / n / 1e9))
PY
  rm -rf $R/gpurun_out/nt_$c$tag
done; done
