#!/bin/bash
# GPU box: SQ / LDS hardware counters of the persistent denominator kernel (VERDICT r4 #1a): which of {issue, LDS bank
# conflicts, LDS latency} bounds the gather passes.  Counter passes are --pmc + --kernel-trace only (8 SQ slots per pass).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/r05_counters_avail.txt 2>&1
python - <<PY > /tmp/passes.txt
import re
avail = open("$R/gpurun_out/r05_counters_avail.txt").read()
want = ["SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_ADDR_CONFLICT","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_WAIT_INST_ANY","SQ_BUSY_CYCLES",
        "SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_LDS_IDX_ACTIVE","SQ_INSTS_SALU","SQ_INSTS_SMEM","SQ_INSTS_VMEM",
        "SQ_LDS_UNALIGNED_STALL","SQ_LDS_MEM_VIOLATIONS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_VMEM","SQ_WAVES","SQ_INST_CYCLES_VMEM",
        "SQ_LDS_DATA_FIFO_FULL","SQ_LDS_CMD_FIFO_FULL","SQ_LDS_ATOMIC_RETURN","SQ_INSTS_FLAT_LDS_ONLY","SQ_INSTS_LDS_DMA"]
have = [c for c in want if re.search(r"\b%s\b" % c, avail)]
for i in range(0, len(have), 8):
    print(" ".join(have[i:i+8]))
PY
cat /tmp/passes.txt
n=0
while read -r line; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $line --kernel-trace -d $R/gpurun_out/pmc_sq$n -o den -- python $R/bench.py --den-only > $R/gpurun_out/pmc_sq$n.log 2>&1
  echo "pmc pass $n ($line) exit $?" >> $R/gpurun_out/summary.txt
done < /tmp/passes.txt
cd $R
{
echo "# rocprofv3 --pmc (SQ counters, 8 per pass, --kernel-trace only) on: python bench.py --den-only  (4 sequences of 589/410/377/502 frames, 30k-state / 1M-arc chain den graph)"
echo "# rows: per kernel and counter, summed over dispatches (rocprofv3 reports SQ counters per XCD/SE row: the sum over the rows of a dispatch)"
for d in gpurun_out/pmc_sq*/; do
  [ -f $d/den_results.db ] && python tools/pmc_stats.py $d/den_results.db 60
done
} > gpurun_out/r05_den_sq_pmc.txt 2>&1
grep -h '{"bound' gpurun_out/pmc_sq1.log | cut -c1-300
rm -rf gpurun_out/pmc_sq*/
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_start.json 2> gpurun_out/bench_start.err
cat gpurun_out/summary.txt; head -50 gpurun_out/r05_den_sq_pmc.txt; cut -c1-400 gpurun_out/r05_bench_start.json
