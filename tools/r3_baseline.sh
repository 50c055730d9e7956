#!/bin/bash
# round 3, first GPU call: the driver's own bench arguments, plain and under a HIP-API + kernel trace (where do the 63 us/step go?)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3_base
mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err && echo "bench20 ok"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20b.json 2>> $O/bench_20.err && echo "bench20b ok"
python3 bench.py --no-cpu-baseline > $O/bench_200.json 2>> $O/bench_20.err && echo "bench200 ok"
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --marker-trace -d $O/trace -o t --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1 && echo "trace ok"
python3 tools/bench_passes.py > $O/passes.txt 2>&1 && echo "passes ok"
cat $O/bench_20.json $O/bench_20b.json $O/bench_200.json
