#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for ls in 0 -1 -2 -3 -4 0; do
  echo "== fwd loop sleep $ls"
  PK2_SEQ_FWD_LOOPSLEEP=$ls timeout 100 python bench.py --lstm-only 2>&1 | grep "^lstm rec" | head -2
done
} > gpurun_out/seq3.txt 2>&1
cat gpurun_out/seq3.txt
