"""How much kernel time of a rocprofv3 kernel trace (.db) ran CONCURRENTLY with other kernels (side-stream experiments):
per kernel name, calls, total time, and the part of it during which another kernel was also running.
Usage: overlap_stats.py results.db [top_n]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ev = []
for i, (n, s, e) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active, last = set(), None
over = defaultdict(float)
for t, d, i in ev:
    if last is not None and len(active) > 1:
        for j in active:
            over[j] += t - last
    last = t
    if d == 1:
        active.add(i)
    else:
        active.discard(i)
agg = defaultdict(lambda: [0, 0.0, 0.0])
for i, (n, s, e) in enumerate(rows):
    a = agg[n[:60]]
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += over[i] / 1e3
span = (max(r[2] for r in rows) - rows[0][1]) / 1e6
print("span %.1f ms, %d kernels, sum of kernel time %.1f ms, concurrent part %.1f ms" %
      (span, len(rows), sum(a[1] for a in agg.values()) / 1e3, sum(a[2] for a in agg.values()) / 1e3))
print("%-62s %6s %10s %12s" % ("kernel", "calls", "total_us", "concurrent_us"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-62s %6d %10.0f %12.0f" % (n, a[0], a[1], a[2]))
