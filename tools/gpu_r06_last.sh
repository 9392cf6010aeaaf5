#!/bin/bash
# GPU box, last session of round 6: the traces of the workloads this session touched (LF-MMI step for the GEMM epilogue change,
# CE, lattice-MMI, TransformerAM), then the default bench line and the denominator-only line.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RD=r06
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_bench.log 2>&1
for w in ce se transformer; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --$w --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$w.log 2>&1
  echo "rocprof $w exit $?"
done
cd $R
python tools/prof_stats.py gpurun_out/prof/bench_results.db 30 > gpurun_out/${RD}_bench_kernel_stats.txt
python tools/step_sequence.py gpurun_out/prof/bench_results.db > gpurun_out/${RD}_step_sequence.txt
python tools/prof_stats.py gpurun_out/prof_ce/ce_results.db 16 > gpurun_out/${RD}_ce_kernel_stats.txt
python tools/prof_stats.py gpurun_out/prof_se/se_results.db 16 > gpurun_out/${RD}_se_kernel_stats.txt
python tools/prof_stats.py gpurun_out/prof_transformer/transformer_results.db 20 > gpurun_out/${RD}_transformer_kernel_stats.txt
python tools/step_sequence.py gpurun_out/prof_transformer/transformer_results.db | tail -1 >> gpurun_out/${RD}_transformer_kernel_stats.txt
rm -rf gpurun_out/prof gpurun_out/prof_ce gpurun_out/prof_se gpurun_out/prof_transformer
tail -2 gpurun_out/${RD}_step_sequence.txt; tail -1 gpurun_out/${RD}_transformer_kernel_stats.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${RD}_bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
timeout 300 python bench.py --den-only 2>/dev/null | tail -1 > gpurun_out/${RD}_den_only.json
python - <<'PY'
import json
b = json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
d = json.loads(open('gpurun_out/r06_den_only.json').read())
print('bench', b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['traffic'], b['roofline']['ms_per_launch'], 'den-only', d['traffic'], d['ms_per_launch'], d['us_per_frame'])
print('input projection', b['roofline_lstm']['input_projection_gemm']['ms_per_launch'], 'f32 window', b['f32_gemm_window']['ms_per_step'], 'allocs', b.get('device_allocs_in_timed_region'))
print(json.dumps(b['secondary_summary']))
PY
