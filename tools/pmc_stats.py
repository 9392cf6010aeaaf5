"""Per-kernel counter totals from a rocprofv3 --pmc results .db (counter value summed over dispatches)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value) from pmc_events "
                   "group by name, counter_name order by 4 desc").fetchall()
print("%-60s %-12s %8s %16s %14s" % ("kernel", "counter", "calls", "sum", "avg/dispatch"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print("%-60s %-12s %8d %16.0f %14.1f" % (r[0][:60], r[1], r[2], r[3], r[4]))
