#!/bin/bash
# two fresh processes (started before anything touches the GPU), one GPU
D=$(mktemp -d)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 5 60 tools/experiments/ipc_probe 0 $D > $D/out0 2>&1 &
P0=$!
timeout -k 5 60 tools/experiments/ipc_probe 1 $D > $D/out1 2>&1 &
P1=$!
wait $P0; echo "rank0 rc=$?"; wait $P1; echo "rank1 rc=$?"
cat $D/out0 $D/out1
