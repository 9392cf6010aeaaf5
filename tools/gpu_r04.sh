#!/bin/bash
# GPU box: everything the round-4 artefacts under profiles/ come from, in one job (ROUND=r04 tools/gpu_profile.sh does the
# bench line, kernel stats, denominator PMC passes and the secondary workloads; this adds the phase timers of the
# persistent kernels -- profile builds libpk2hip_{seqp,dpp,latp,bgp}.so -- and the denominator S x A sweep).
mkdir -p gpurun_out; export TMPDIR=/tmp
ROUND=r04 bash tools/gpu_profile.sh > gpurun_out/r04_profile.log 2>&1
{
echo "# in-kernel phase timers of the small-batch persistent recurrences (lstm_persist_seq.hip; profile build = -DPK2_SEQ_PROFILE), bench.py --lstm-only: one layer, T = 589, B = 4, both directions; shader clocks per step, wave 0 of rank 0 of a team"
echo "# round-3 kernels (PK2_LSTM_SEQ_FORM=1)"
PK2_LSTM_SEQ_FORM=1 PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_seqp.so timeout 300 python bench.py --lstm-only > /tmp/o.txt 2>&1
grep "^lstm_fwd_seq " /tmp/o.txt | grep "589 steps" | head -2; grep "^lstm_bwd_seq " /tmp/o.txt | grep "589 steps" | head -2
echo "# round-4 kernels (default)"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_seqp.so timeout 300 python bench.py --lstm-only > /tmp/o.txt 2>&1
grep "^lstm_fwd_seq2" /tmp/o.txt | grep "589 steps" | head -2; grep "^lstm_bwd_seq2" /tmp/o.txt | grep "589 steps" | head -2
echo "# the same launches without timers (us per step over the whole launch): round-3 kernels, round-4 kernels, round-4 kernels with agent-scope hand-over stores (PK2_SEQ_STORE_MODE=0)"
PK2_LSTM_SEQ_FORM=1 timeout 300 python bench.py --lstm-only 2>/dev/null | grep "^lstm rec" | head -1
timeout 300 python bench.py --lstm-only 2>/dev/null | grep "^lstm rec" | head -1
PK2_SEQ_STORE_MODE=0 timeout 300 python bench.py --lstm-only 2>/dev/null | grep "^lstm rec" | head -1
} > gpurun_out/r04_seq_phases.txt 2>&1
{
echo "# in-kernel phase timers of the other persistent kernels (profile builds), one MI355X"
echo "# persistent denominator, bench.py --den-only, -DPK2_DP_PROFILE: shader clocks per frame (summed over the rank-0 workgroups of the four recursions of a direction: ratios, not absolute times) and the 10 ns timeline of frame 100 on ranks 0..3"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_dpp.so timeout 300 python bench.py --den-only > /tmp/o.txt 2>&1
grep "^den_persist2" /tmp/o.txt | tail -2; grep -A5 "^timeline fwd" /tmp/o.txt | tail -6 | head -6; grep -A5 "^timeline bwd" /tmp/o.txt | tail -6 | head -6
echo "# lattice decoder, bench.py --se (8 utterances), -DPK2_LATP_PROFILE: 10 ns ticks per frame, ranks 0 and 1 of the team of utterance 0"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_latp.so timeout 300 python bench.py --se --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep "^kth_below\|^lat_frames_persist" | head -3
echo "# large-batch LSTM recurrences, bench.py --ce (256 x 80), -DPK2_BIG_PROFILE: 10 ns ticks per step, thread 0 of rank 0 of a team"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_bgp.so timeout 300 python bench.py --ce --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^lstm_.wd_big" > /tmp/big.txt
grep "^lstm_fwd_big" /tmp/big.txt | head -2; grep "^lstm_bwd_big" /tmp/big.txt | head -2
} > gpurun_out/r04_persistent_phases.txt 2>&1
SWEEP_S="10000 30000 40000 50000 65000" SWEEP_A="500000 1000000 1500000 2000000" bash tools/gpu_den_sweep.sh > /dev/null 2>&1
cp gpurun_out/den_sweep.txt gpurun_out/r04_den_sweep.txt
tail -3 gpurun_out/summary.txt; wc -l gpurun_out/r04_*.txt; head -c 600 gpurun_out/r04_bench.json
