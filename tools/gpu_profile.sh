#!/bin/bash
# GPU box: produce the artefacts kept under profiles/ (kernel-trace stats of the bench command, HBM counters of
# the denominator call in separate --pmc passes) plus the bench JSON line.  ROUND=r02 names the files.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RD=${ROUND:-r02}
timeout 900 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 3 > gpurun_out/${RD}_bench.json 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
echo "rocprof bench exit $?" >> $R/gpurun_out/summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o den -- python $R/bench.py --den-only > $R/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c exit $?" >> $R/gpurun_out/summary.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/den_trace -o den -- python $R/bench.py --den-only > $R/gpurun_out/den_trace.log 2>&1
if [ -z "$SKIP_SECONDARY" ]; then
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_mfma -o gemm -- python $R/bench.py --gemm-only > $R/gpurun_out/pmc_mfma.log 2>&1
  echo "pmc mfma exit $?" >> $R/gpurun_out/summary.txt
  for w in ce se transformer; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --$w --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$w.log 2>&1
    echo "rocprof $w exit $?" >> $R/gpurun_out/summary.txt
  done
fi
cd $R
python tools/prof_stats.py gpurun_out/prof/bench_results.db 30 > gpurun_out/${RD}_bench_kernel_stats.txt
python tools/prof_stats.py gpurun_out/den_trace/den_results.db 12 > gpurun_out/${RD}_den_kernel_stats.txt
grep -o '{"bound.*' gpurun_out/den_trace.log > gpurun_out/${RD}_den_only.json
python tools/pmc_stats.py gpurun_out/pmc_FETCH_SIZE/den_results.db > gpurun_out/${RD}_den_pmc.txt
python tools/pmc_stats.py gpurun_out/pmc_WRITE_SIZE/den_results.db >> gpurun_out/${RD}_den_pmc.txt
python tools/den_traffic.py gpurun_out/pmc_FETCH_SIZE/den_results.db gpurun_out/pmc_WRITE_SIZE/den_results.db gpurun_out/${RD}_den_only.json gpurun_out/${RD}_den_traffic.json
if [ -z "$SKIP_SECONDARY" ]; then
  python tools/prof_stats.py gpurun_out/prof_ce/ce_results.db 16 > gpurun_out/${RD}_ce_kernel_stats.txt
  python tools/prof_stats.py gpurun_out/prof_se/se_results.db 16 > gpurun_out/${RD}_se_kernel_stats.txt
  python tools/prof_stats.py gpurun_out/prof_transformer/transformer_results.db 20 > gpurun_out/${RD}_transformer_kernel_stats.txt
  grep -h '"metric"' gpurun_out/prof_ce.log gpurun_out/prof_se.log gpurun_out/prof_transformer.log > gpurun_out/${RD}_secondary_bench.json
  python tools/pmc_stats.py gpurun_out/pmc_mfma/gemm_results.db > gpurun_out/${RD}_gemm_mfma_pmc.txt
  grep gemm gpurun_out/pmc_mfma.log >> gpurun_out/${RD}_gemm_mfma_pmc.txt
fi
rm -rf gpurun_out/prof gpurun_out/prof_ce gpurun_out/prof_se gpurun_out/prof_transformer gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_mfma gpurun_out/den_trace
cat gpurun_out/summary.txt; cat gpurun_out/${RD}_bench.json
