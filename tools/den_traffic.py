"""HBM traffic of one denominator call from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --den-only`:
sums the counters over every kernel of the command and divides by the number of denominator calls (launches of its tail kernel).
Writes the JSON bench.py reads for `roofline.traffic`.  Usage: den_traffic.py fetch.db write.db den_only.json out.json"""
import json
import sqlite3
import sys


def total(db, counter):
    cur = sqlite3.connect(db).cursor()
    s = cur.execute("select sum(counter_value) from pmc_events where counter_name = ?", (counter,)).fetchone()[0]
    # one launch of the call's tail kernel per denominator call (round 6: den_tail1; before: den_scales)
    calls = 0
    for pat in ("%den_tail1%", "%den_scales%"):
        calls = cur.execute("select count(*) from pmc_events where counter_name = ? and name like ?", (counter, pat)).fetchone()[0]
        if calls:
            break
    # (rocprofv3 reports one row per dispatch for the TCC-derived counters)
    return float(s), int(calls)


fetch_kb, calls_f = total(sys.argv[1], "FETCH_SIZE")
write_kb, calls_w = total(sys.argv[2], "WRITE_SIZE")
den = json.load(open(sys.argv[3]))
import re
m = re.search(r"\[(\d+), (\d+), (\d+), (\d+)\] frames, (\w+) den graph \((\d+) states, (\d+) arcs", den["workload"])
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) on `python bench.py --den-only`, "
              "%d calls, all kernels of the call (tools/gpu_profile.sh)" % calls_f,
    "lengths": [int(m.group(i)) for i in range(1, 5)], "topology": m.group(5), "states": int(m.group(6)), "arcs": int(m.group(7)),
    "fetch_size_kb_per_call": fetch_kb / max(1, calls_f), "write_size_kb_per_call": write_kb / max(1, calls_w),
    "traffic_bytes_raw": 1024.0 * (fetch_kb / max(1, calls_f) + write_kb / max(1, calls_w)),
    "traffic_bytes_fetch_x2": 1024.0 * (2.0 * fetch_kb / max(1, calls_f) + write_kb / max(1, calls_w)),
    "note": "gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md HBM section); the x2 figure "
            "is an upper bound because the 16/32-byte gathers are not wide reads",
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
