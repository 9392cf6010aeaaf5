#!/bin/bash
# GPU box: what the gradient exchange of an 8-rank job would do to a step, priced on ONE GPU.  One rank with a
# process group (PK2_HVD_SINGLE_RANK_GROUP=1: the RCCL path of hvd.py runs, its all-reduce of one rank is a copy) and, in front
# of every all-reduce call, the stand-in kernel of PK2_HVD_FAKE_PEER=blocks,passes on the same stream: `blocks` workgroups
# streaming a reduce-copy over the bucket `passes` times.  Both schedules (PK2_HVD_OVERLAP=0: one all-reduce behind backward on
# the compute stream; 1: per bucket on a side stream under backward), 200 steps each; then the tests that came with it.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
run() { env "$@" PK2_HVD_SINGLE_RANK_GROUP=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '|', d['ms_per_step'], d.get('breakdown_ms'), 'exchange_ms', d['exchange']['exchange_ms'], d['exchange']['schedule'], d['persistent_health'])"; }
run PK2_HVD_OVERLAP=0
run PK2_HVD_OVERLAP=1
for spec in 16,4 16,12 32,4 32,12 64,12; do
run PK2_HVD_OVERLAP=0 PK2_HVD_FAKE_PEER=$spec
run PK2_HVD_OVERLAP=1 PK2_HVD_FAKE_PEER=$spec
done
} > gpurun_out/r05_peer.txt 2>&1
cat gpurun_out/r05_peer.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_comm.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
