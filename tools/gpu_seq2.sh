#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend_nn.py -m gpu -q -x --tb=short -p no:cacheprovider -k "persistent_recurrence or blstm_3x512 or lstmam_matches" > gpurun_out/seq_tests.log 2>&1
echo "seq tests exit $?"; tail -5 gpurun_out/seq_tests.log | cut -c1-300
{
for f in 1 2; do
  echo "== form $f lstm-only"
  PK2_LSTM_SEQ_FORM=$f timeout 300 python bench.py --lstm-only 2>/dev/null | grep "^lstm rec"
done
echo "== profile (form 2)"
PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_seqp.so timeout 300 python bench.py --lstm-only > /tmp/o.txt 2>&1
grep "^lstm_fwd_seq2" /tmp/o.txt | grep "589 steps" | head -3
grep "^lstm_bwd_seq2" /tmp/o.txt | grep "589 steps" | head -3
for f in 1 2 1 2; do
  echo "== PK2_LSTM_SEQ_FORM=$f bench"
  PK2_LSTM_SEQ_FORM=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'), d.get('parity'))"
done
} > gpurun_out/seq_ab2.txt 2>&1
cat gpurun_out/seq_ab2.txt | cut -c1-420
