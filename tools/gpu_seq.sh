#!/bin/bash
# GPU box: the small-batch persistent recurrences (lstm_persist_seq.hip): parity of the default form, us/step of both forms,
# in-kernel phase timers (profile build libpk2hip_seqp.so = -DPK2_SEQ_PROFILE), bench A/B.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend_nn.py -m gpu -q -x --tb=short -p no:cacheprovider -k "persistent_recurrence or blstm_3x512 or lstmam_matches" > gpurun_out/seq_tests.log 2>&1
echo "seq tests exit $?"; tail -5 gpurun_out/seq_tests.log | cut -c1-300
{
for f in 1 2; do
  echo "== PK2_LSTM_SEQ_FORM=$f lstm-only"
  PK2_LSTM_SEQ_FORM=$f timeout 300 python bench.py --lstm-only 2>/dev/null | grep "^lstm"
done
for f in 1 2; do
  echo "== PK2_LSTM_SEQ_FORM=$f phase timers (profile build)"
  PK2_LIB=$PWD/pykaldi2_amd/libpk2hip_seqp.so PK2_LSTM_SEQ_FORM=$f timeout 300 python bench.py --lstm-only 2>/dev/null | grep "^lstm_.wd_seq" | sort | uniq -c | sort -rn | head -12
done
for f in 1 2 1 2; do
  echo "== PK2_LSTM_SEQ_FORM=$f bench"
  PK2_LSTM_SEQ_FORM=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'))"
done
} > gpurun_out/seq_ab.txt 2>&1
cat gpurun_out/seq_ab.txt | cut -c1-400
