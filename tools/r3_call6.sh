#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3_call6
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_comm_gpu.py tests/test_shard_chain.py tests/test_probes.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
