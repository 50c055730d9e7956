#!/bin/bash
# the N = 2 control flow of bench.py on the one-GPU box: two ranks on cuda:0, gloo, direct exchange (3 processes on the GPU at most)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3_rehearse; mkdir -p $O
i=0
for w in "" "--workload 4k_deferred_gi" "--one-work-stream"; do
i=$((i+1))
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2963$i bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu $w > $O/out$i.json 2> $O/err$i.log; echo "rc=$? ($w)"; cat $O/out$i.json; grep -v "^\[W\|Gloo\|^$" $O/err$i.log | tail -12
done
