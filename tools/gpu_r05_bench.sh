#!/bin/bash
# GPU box: the default bench line with its secondary configurations, plus the tests added this round.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_gpu_comm.py "tests/test_gpu_chain.py::test_alpha_beta_guard_product_and_both_oracles_on_the_same_input" tests/test_gpu_chain.py::test_alpha_beta_guard_follows_kaldis_rule -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -15
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_full.json 2> gpurun_out/bench_full.err ) 2>&1 | grep real
tail -12 gpurun_out/bench_full.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["us_per_frame"], d["parity"]["ok"], d["cpu_baseline"]["value"])
for k,v in d.get("secondary",{}).items():
    print(k, json.dumps(v)[:900])
PY
} > gpurun_out/r05_bench_full.txt 2>&1
cat gpurun_out/r05_bench_full.txt | cut -c1-1000
