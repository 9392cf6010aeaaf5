#!/bin/bash
# GPU box: in-kernel timers of the f32 GEMM tiles (profile build, -DPK2_GEMM_PROFILE) on one shape: start tick, main loop, epilogue per tile.
mkdir -p gpurun_out; export TMPDIR=/tmp
SH="${GEMM_SHAPE:-0,1,2356,512,512}"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_gprof.so timeout 300 python bench.py --gemm-only --gemm-shapes "$SH" > gpurun_out/gemm_prof.txt 2>&1
grep -c "gemm tile" gpurun_out/gemm_prof.txt
python - <<'PY'
import re
rows=[]
for l in open('gpurun_out/gemm_prof.txt'):
    m=re.match(r"gemm tile \((\d+),(\d+)\) of grid \((\d+),(\d+)\) tiles=(\d) xcc (\d): start (\d+) main loop (\d+) epilogue (\d+)", l)
    if m: rows.append(tuple(int(v) for v in m.groups()))
    elif l.startswith("gemm ta"): print(l.strip())
# group launches by start-tick gaps
rows.sort(key=lambda r: r[6])
launches=[]; cur=[]
for r in rows:
    if cur and r[6]-cur[-1][6] > 1500: launches.append(cur); cur=[]
    cur.append(r)
if cur: launches.append(cur)
print(len(launches), "launches")
for L in launches[-3:]:
    t0=min(r[6] for r in L)
    print("launch: tiles %d, start spread %d ticks, end of last tile %d ticks after first start; main loop min/avg/max %d/%d/%d, epilogue avg %d" % (
        len(L), max(r[6] for r in L)-t0, max(r[6]+r[7]+r[8] for r in L)-t0, min(r[7] for r in L), sum(r[7] for r in L)/len(L), max(r[7] for r in L), sum(r[8] for r in L)/len(L)))
    for r in L[:12]: print("   tile", r[:2], "xcc", r[5], "start +%d" % (r[6]-t0), "loop", r[7], "epi", r[8])
PY
