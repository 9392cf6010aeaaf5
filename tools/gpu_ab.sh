#!/bin/bash
# A/B of an environment switch on the bench: AB_VAR=NAME AB_VALS="0 1" [AB_ARGS="--steps 20 --warmup 8"]
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
  for v in $AB_VALS; do
    env $AB_VAR=$v timeout 300 python bench.py ${AB_ARGS:---steps 20 --warmup 8} --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
    python - "$AB_VAR=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("breakdown_ms"))
PY
  done
done
