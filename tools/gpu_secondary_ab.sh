#!/bin/bash
# secondary workloads under both GEMM arithmetic paths (same box)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for w in --ce --transformer --se; do
  for ar in f32 bf16x3; do
    PK2_GEMM_ARITH=$ar timeout 600 python bench.py $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/s.json
    python - "$w $ar" <<'PY'
import json, sys
d = json.load(open("/tmp/s.json"))
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "value", d["value"], "parity", (d.get("parity") or {}).get("ok"), {k: v for k, v in (d.get("breakdown_ms") or {}).items()})
PY
  done
done
