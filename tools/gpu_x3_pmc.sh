#!/bin/bash
# SQ counters of the bf16x3 GEMM on one shape (separate --pmc passes, kernel-trace only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp
SHAPE=${X3_SHAPE:-0 1 20480 4096 1024}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/x3pmc_$i -o x3 -- python $R/tools/dbg/gemm_one.py $SHAPE > $R/gpurun_out/x3pmc_$i.log 2>&1
  echo "pass $i exit $?"
done
cd $R
python tools/pmc_stats.py gpurun_out/x3pmc_1 gpurun_out/x3pmc_2 gpurun_out/x3pmc_3 2>&1 | grep -i "gemm" | head -60 | tee gpurun_out/x3_pmc.txt
