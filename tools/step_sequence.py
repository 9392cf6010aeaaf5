"""Kernel sequence of the LAST optimiser step in a rocprofv3 results .db (kernels between the last two adam_kernel
launches): name, duration, gap to the previous kernel's end.  Usage: step_sequence.py results.db [steps back from the last, default 0]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[0] or "sgd_kernel" in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lo, hi = (marks[-2 - back] + 1, marks[-1 - back] + 1) if len(marks) >= 2 + back else (0, len(rows))
prev = rows[lo - 1][2] if lo > 0 else rows[lo][1]
small = 0
for name, st, en in rows[lo:hi]:
    dur = (en - st) / 1e3
    print("%9.2f us  gap %8.2f  %s" % (dur, (st - prev) / 1e3, name[:100]))
    prev = en
    small += dur < 40.0
print("# %d launches in the step, %d of them shorter than 40 us; span %.3f ms" % (hi - lo, small, (rows[hi - 1][2] - rows[lo][1]) / 1e6))
