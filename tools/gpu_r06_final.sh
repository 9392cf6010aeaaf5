#!/bin/bash
# GPU box, end of round 6.  PHASE=1: the counter / trace passes (PMC traffic of the denominator call first: bench.py quotes the
# committed file).  PHASE=2 (after profiles/r06_den_traffic.json has been committed): the bench lines that quote it.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RD=r06
if [ "${PHASE:-1}" = "1" ]; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o den -- python $R/bench.py --den-only > $R/gpurun_out/pmc_$c.log 2>&1
    echo "pmc $c exit $?"
  done
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/den_trace -o den -- python $R/bench.py --den-only > $R/gpurun_out/den_trace.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_bench.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_mfma -o gemm -- python $R/bench.py --gemm-only > $R/gpurun_out/pmc_mfma.log 2>&1
  for w in ce se transformer; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --$w --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$w.log 2>&1
    echo "rocprof $w exit $?"
  done
  cd $R
  python tools/prof_stats.py gpurun_out/prof/bench_results.db 30 > gpurun_out/${RD}_bench_kernel_stats.txt
  python tools/step_sequence.py gpurun_out/prof/bench_results.db > gpurun_out/${RD}_step_sequence.txt
  python tools/prof_stats.py gpurun_out/den_trace/den_results.db 12 > gpurun_out/${RD}_den_kernel_stats.txt
  grep -o '{"bound.*' gpurun_out/den_trace.log > gpurun_out/${RD}_den_only_profiled.json
  python tools/pmc_stats.py gpurun_out/pmc_FETCH_SIZE/den_results.db > gpurun_out/${RD}_den_pmc.txt
  python tools/pmc_stats.py gpurun_out/pmc_WRITE_SIZE/den_results.db >> gpurun_out/${RD}_den_pmc.txt
  python tools/den_traffic.py gpurun_out/pmc_FETCH_SIZE/den_results.db gpurun_out/pmc_WRITE_SIZE/den_results.db gpurun_out/${RD}_den_only_profiled.json gpurun_out/${RD}_den_traffic.json
  python tools/prof_stats.py gpurun_out/prof_ce/ce_results.db 16 > gpurun_out/${RD}_ce_kernel_stats.txt
  python tools/prof_stats.py gpurun_out/prof_se/se_results.db 16 > gpurun_out/${RD}_se_kernel_stats.txt
  python tools/prof_stats.py gpurun_out/prof_transformer/transformer_results.db 20 > gpurun_out/${RD}_transformer_kernel_stats.txt
  python tools/step_sequence.py gpurun_out/prof_transformer/transformer_results.db | tail -1 >> gpurun_out/${RD}_transformer_kernel_stats.txt
  python tools/pmc_stats.py gpurun_out/pmc_mfma/gemm_results.db > gpurun_out/${RD}_gemm_mfma_pmc.txt
  grep gemm gpurun_out/pmc_mfma.log >> gpurun_out/${RD}_gemm_mfma_pmc.txt
  rm -rf gpurun_out/prof gpurun_out/prof_ce gpurun_out/prof_se gpurun_out/prof_transformer gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_mfma gpurun_out/den_trace
  cat gpurun_out/${RD}_den_traffic.json; head -12 gpurun_out/${RD}_den_kernel_stats.txt; tail -2 gpurun_out/${RD}_step_sequence.txt
else
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${RD}_bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
  timeout 300 python bench.py --den-only 2>/dev/null | tail -1 > gpurun_out/${RD}_den_only.json
  python - <<'PY'
import json
b = json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
d = json.loads(open('gpurun_out/r06_den_only.json').read())
print('bench', b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['traffic'], b['roofline']['ms_per_launch'], 'den-only', d['traffic'], d['ms_per_launch'], d['us_per_frame'])
print(json.dumps(b['secondary_summary']))
PY
fi
