#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3_call2
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_rt.py -x -q -m gpu > $O/test_rt.log 2>&1; echo "test_rt rc=$?"
tail -15 $O/test_rt.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --deselect tests/test_rt.py > $O/test_all.log 2>&1; echo "test_all rc=$?"
tail -5 $O/test_all.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench.err && cat $O/bench_20.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20b.json 2>> $O/bench.err && cat $O/bench_20b.json
