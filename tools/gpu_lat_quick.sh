#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lattice.py -q -x -m gpu -k "take_turns or persistent_decoder_is_in_use or (decoder_matches_oracle and lds)" 2>&1 | tail -3
[ -f pykaldi2_amd/libpk2hip_latp.so ] && PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_latp.so timeout 300 python bench.py --se --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep "^lat_frames_persist rank 0" | head -1
for i in 1 2; do timeout 300 python bench.py --se --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('lattice_ms'))"; done
