"""GPU idle time between consecutive kernels of a rocprofv3 results .db (kernel trace): where the stream is not
busy.  Usage: gap_stats.py results.db [min_gap_us]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
pairs = defaultdict(lambda: [0, 0.0])
busy = sum(r[2] - r[1] for r in rows) / 1e3
span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
tot_gap = 0.0
last_end = rows[0][2]
last_name = rows[0][0]
for name, st, en in rows[1:]:
    g = (st - last_end) / 1e3
    if g > thr:
        k = (last_name[:40], name[:40])
        pairs[k][0] += 1
        pairs[k][1] += g
        tot_gap += g
    if en > last_end:
        last_end, last_name = en, name
print("span %.1f ms, kernel busy (sum) %.1f ms, idle in gaps > %.1f us: %.1f ms" % (span / 1e3, busy / 1e3, thr, tot_gap / 1e3))
print("%-42s %-42s %7s %10s %8s" % ("after kernel", "before kernel", "count", "total_us", "avg_us"))
for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-42s %-42s %7d %10.0f %8.1f" % (k[0], k[1], v[0], v[1], v[1] / v[0]))
