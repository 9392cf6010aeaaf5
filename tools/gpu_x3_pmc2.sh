#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp
for w in 1 0; do for z in "" 1; do
  PK2_ZERO=$z PK2_GEMM_WIDE=$w PYTHONPATH=$R timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/x3w_$w$z -o x3 -- python $R/tools/dbg/gemm_one.py 0 1 20480 4096 1024 > $R/gpurun_out/x3w_$w$z.log 2>&1
  echo "== wide=$w zero=$z"; grep "bf16x3" $R/gpurun_out/x3w_$w$z.log
  python $R/tools/pmc_stats.py $R/gpurun_out/x3w_$w$z/x3_results.db 40 | grep "true>\|wide" 
done; done
