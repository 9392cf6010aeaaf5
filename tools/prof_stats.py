"""Per-kernel summary (calls, total, avg, min, max, %) from a rocprofv3 results .db."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                   "max(end-start)/1e3, max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
t = cur.execute("select min(start), max(end) from kernels").fetchone()
print("total kernel time %.1f us over a %.3f s span" % (tot, (t[1] - t[0]) / 1e9))
print("%-62s %7s %11s %9s %8s %9s %6s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "lds"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-62s %7d %11.0f %9.2f %8.2f %9.2f %5.1f%% %5s %7s" % (r[0][:62], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7]))
