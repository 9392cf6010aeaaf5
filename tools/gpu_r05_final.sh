#!/bin/bash
# GPU box, last job of round 5: the bench lines that quote `roofline.traffic` regenerated AFTER the last PMC traffic measurement
# was committed (profiles/r05_den_traffic.json), so that r05_bench.json, r05_den_only.json, r05_secondary_bench.json and the
# traffic file agree; the secondary workloads' kernel statistics with them.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r05_bench.json 2> gpurun_out/bench.err
timeout 300 python bench.py --den-only 2>/dev/null | tail -1 > gpurun_out/r05_den_only.json
cd /tmp
for w in ce se transformer; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --$w --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$w.log 2>&1
done
cd $R
python tools/prof_stats.py gpurun_out/prof_ce/ce_results.db 16 > gpurun_out/r05_ce_kernel_stats.txt
python tools/prof_stats.py gpurun_out/prof_se/se_results.db 16 > gpurun_out/r05_se_kernel_stats.txt
python tools/prof_stats.py gpurun_out/prof_transformer/transformer_results.db 20 > gpurun_out/r05_transformer_kernel_stats.txt
grep -h '"metric"' gpurun_out/prof_ce.log gpurun_out/prof_se.log gpurun_out/prof_transformer.log > gpurun_out/r05_secondary_bench.json
rm -rf gpurun_out/prof_ce gpurun_out/prof_se gpurun_out/prof_transformer
python - <<'PY'
import json
b = json.loads(open('gpurun_out/r05_bench.json').read().strip().splitlines()[-1])
d = json.loads(open('gpurun_out/r05_den_only.json').read())
t = json.load(open('profiles/r05_den_traffic.json'))
print('bench', b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['traffic'], 'den-only', d['traffic'], 'file', t['traffic_bytes_raw'])
for l in open('gpurun_out/r05_secondary_bench.json'):
    j = json.loads(l); print(j['metric'][:40], j['value'], j['ms_per_step'], (j.get('roofline_denominator') or {}).get('traffic'))
PY
