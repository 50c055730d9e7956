#!/bin/bash
# the sharded chain exchanges mip 1 instead of mip 0: parity of the emulated / two-rank / one-rank-communicator chains, the rank emulation's
# frame times, the two-rank rehearsal on one GPU
set -o pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_shard_chain.py tests/test_comm_gpu.py tests/test_post_gpu.py tests/test_dist_cpu.py -x -q -m gpu > gpurun_out/r4_round12_tests.log 2>&1 || { tail -40 gpurun_out/r4_round12_tests.log; exit 1; }
tail -3 gpurun_out/r4_round12_tests.log
bash tools/experiments/r4/r4_n8.sh
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29651 bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu 2> gpurun_out/r4_rehearse.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('rehearsal N=2 on one GPU: ms/step', d['ms_per_step'], 'sharded_equals_unsharded', c.get('sharded_equals_unsharded'), 'preflight', c.get('preflight'))" || tail -5 gpurun_out/r4_rehearse.err
