#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3_rehearse2; mkdir -p $O
for i in 1 2 3 4; do
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2964$i bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu > $O/out$i.json 2> $O/err$i.log; echo "run $i rc=$?"; grep -i "error\|Traceback\|sah status\|raise\|Exception" $O/err$i.log | head -5
done
