#!/bin/bash
# lattice decoder: parity tests, then the lattice-MMI bench line under the default library and an experiment build, alternating
# (same box, same job):  PK2_BUILD_TAG=<tag> PK2_EXTRA_FLAGS=... python -m pykaldi2_amd.build;  TAG=<tag> bash tools/gpu_lat_ab.sh
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
TAG=${TAG:-nomerge}
[ -z "$SKIP_TESTS" ] && timeout 1500 python -m pytest tests/test_gpu_lattice.py -q -x -m gpu 2>&1 | tail -3
for i in 1 2 3 4 5; do
  for lib in libpk2hip.so libpk2hip_$TAG.so; do
    PK2_LIB=$PWD/pykaldi2_amd/$lib timeout 600 python bench.py --se --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'], d['roofline'].get('us_per_frame'), d.get('lattice_ms'))"
  done
done
