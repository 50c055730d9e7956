#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3_call3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_rt.py tests/test_golden.py tests/test_host_facade_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 600 python3 bench.py --workload 4k_probe_gi_chain_traced --steps 10 --warmup 2 --no-cpu-baseline > $O/traced.json 2> $O/traced.err; echo "traced rc=$?"; cat $O/traced.json; tail -3 $O/traced.err
timeout -k 10 300 python3 bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 5 --no-cpu-baseline > $O/chain.json 2>> $O/traced.err; cat $O/chain.json
