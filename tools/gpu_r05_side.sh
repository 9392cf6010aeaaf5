#!/bin/bash
# GPU box: the weight-gradient products of the backward pass on a side stream, unmasked and confined to k CUs per XCD
# (VERDICT r4 #2), against the default (everything on one stream).  Two bench runs per variant.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
python tools/dbg/cu_mask_probe.py 2>&1 | tail -5
run() { for i in 1 2; do env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d.get('breakdown_ms'), d['persistent_health']['lstm_persist_abort'])"; done; }
run PK2_NOP=1
run PK2_SIDE_STREAM=1
run PK2_SIDE_STREAM=1 PK2_SIDE_CU_PER_XCD=4
run PK2_SIDE_STREAM=1 PK2_SIDE_CU_PER_XCD=8
run PK2_SIDE_STREAM=1 PK2_SIDE_CU_PER_XCD=16
run PK2_SIDE_STREAM=1 PK2_SIDE_CU_PER_XCD=24
} > gpurun_out/r05_side_stream.txt 2>&1
cat gpurun_out/r05_side_stream.txt | cut -c1-300
{
for w in se transformer; do
timeout 600 python bench.py --$w --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], json.dumps(d['parity'])[:900], json.dumps(d['cpu_baseline'])[:300])"
done
} > gpurun_out/r05_sec_check.txt 2>&1
cat gpurun_out/r05_sec_check.txt
