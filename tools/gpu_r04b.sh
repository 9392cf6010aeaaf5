#!/bin/bash
# GPU box: refresh of the round-4 artefacts after the GEMM / denominator changes late in the round: bench line, kernel stats,
# denominator PMC passes, secondary workloads (tools/gpu_profile.sh), the denominator's frame timeline (profile build
# libpk2hip_dpp.so), the GEMM table (tools/gpu_gemm_tr.sh) and the S x A sweep.
mkdir -p gpurun_out; export TMPDIR=/tmp
ROUND=r04 bash tools/gpu_profile.sh > gpurun_out/r04_profile.log 2>&1
{
echo "# persistent denominator, bench.py --den-only, -DPK2_DP_PROFILE: shader clocks per frame (summed over the rank-0 workgroups of the four recursions of a direction: ratios, not absolute times) and the 10 ns timeline of frame 100 on all 32 ranks"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so timeout 300 python bench.py --den-only > /tmp/o.txt 2>&1
grep "^den_persist2" /tmp/o.txt | tail -2; grep -A32 "^timeline fwd" /tmp/o.txt | tail -33; grep -A32 "^timeline bwd" /tmp/o.txt | tail -33
} > gpurun_out/r04_den_timeline.txt 2>&1
bash tools/gpu_gemm_tr.sh > /dev/null 2>&1; cp gpurun_out/gemm_tr.txt gpurun_out/r04_gemm_shapes.txt
SWEEP_S="10000 30000 36000 40000 45000 50000 55000 65000" SWEEP_A="500000 1000000 1500000 2000000" SWEEP_MODES=default bash tools/gpu_den_sweep.sh > /dev/null 2>&1
cp gpurun_out/den_sweep.txt gpurun_out/r04_den_sweep.txt
tail -3 gpurun_out/summary.txt; wc -l gpurun_out/r04_*.txt; head -c 900 gpurun_out/r04_bench.json
