#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_frontend_nn.py -m gpu -q -x --tb=short -p no:cacheprovider -k "persistent_recurrence or blstm_3x512 or lstmam_matches" 2>&1 | tail -2
timeout 300 python bench.py --lstm-only 2>/dev/null | grep "^lstm rec"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_seqp.so timeout 300 python bench.py --lstm-only > /tmp/o.txt 2>&1
grep "^lstm_fwd_seq2" /tmp/o.txt | grep "589 steps" | head -2
grep "^lstm_bwd_seq2" /tmp/o.txt | grep "589 steps" | head -3
} > gpurun_out/seq3.txt 2>&1
cat gpurun_out/seq3.txt | cut -c1-100,300-600
