#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_transformer.py tests/test_gpu_comm.py tests/test_gpu_chain.py -q -x -m gpu --tb=short -p no:cacheprovider -k "transformer or guard or peaked or bench_size" -s 2>&1 | grep -v amdgpu.ids | tail -15
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["dtype"])
print(json.dumps(d)[-1500:])
PY
