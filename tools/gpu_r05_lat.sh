#!/bin/bash
# GPU box: lattice path after a change -- its parity tests, the --se line, kernel stats, and (libpk2hip_find.so) the phase
# timers of the pruning pass.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
timeout 1200 python -m pytest tests/test_gpu_lattice.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do
timeout 600 python bench.py --se --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('se', d['value'], d['ms_per_step'], d['lattice_ms'], d['roofline']['us_per_frame'])"
done
if [ -f pykaldi2_amd/libpk2hip_find.so ]; then
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_find.so timeout 600 python bench.py --se --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep "^finish utt" | head -8
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_se -o se -- python $R/bench.py --se --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_se.log 2>&1
cd $R
python tools/prof_stats.py gpurun_out/prof_se/se_results.db 14
rm -rf gpurun_out/prof_se
} > gpurun_out/r05_lat.txt 2>&1
cat gpurun_out/r05_lat.txt | cut -c1-250
