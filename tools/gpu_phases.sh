export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "# in-kernel phase timers of the round-3 persistent kernels (profile builds: PK2_BUILD_TAG=... PK2_EXTRA_FLAGS=-D... python -m pykaldi2_amd.build; PK2_LIB selects them), one MI355X"
echo "# lattice decoder, bench.py --se (8 utterances), -DPK2_LATP_PROFILE: 10 ns ticks per frame, ranks 0 and 1 of the team of utterance 0"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_latp.so timeout 300 python bench.py --se --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep "^kth_below\|^lat_frames_persist" | head -3
echo "# large-batch LSTM recurrences, bench.py --ce (256 x 80), -DPK2_BIG_PROFILE: 10 ns ticks per step, thread 0 of rank 0 of a team"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_bgp.so timeout 300 python bench.py --ce --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^lstm_.wd_big" > /tmp/big.txt
grep "^lstm_fwd_big" /tmp/big.txt | head -2; grep "^lstm_bwd_big" /tmp/big.txt | head -2
} > gpurun_out/r03_persistent_phases.txt
cat gpurun_out/r03_persistent_phases.txt | cut -c1-300
