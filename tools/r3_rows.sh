#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_rt.py tests/test_lighting_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29653 bench.py --gpus 2 --steps 6 --warmup 2 --rehearse-on-one-gpu --workload 4k_probe_gi_chain_traced --no-cpu-baseline > gpurun_out/r3_rehearse_traced.json 2> gpurun_out/r3_rehearse_traced.err || { tail -20 gpurun_out/r3_rehearse_traced.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r3_rehearse_traced.json')); print(d['ms_per_step'], d['config']['parallelism'][:60], d['config']['traced']['rtao_ms'])"
python bench.py --workload 4k_probe_gi_chain_traced --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('one gpu', d['ms_per_step'])"
