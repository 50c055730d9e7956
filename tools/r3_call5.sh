#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3_call5
mkdir -p $O
bash tools/r3_call4.sh 2>&1 | grep -v histogram | tail -5
timeout -k 10 900 python3 -m pytest tests/test_lighting_gpu.py tests/test_golden.py tests/test_post_gpu.py tests/test_lighting_ext_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for a in "" "--repack-lpv"; do python3 bench.py --no-cpu-baseline $a 2>>$O/err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench $a', d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['frac'])"; done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench 20', d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['frac'])"
python3 tools/bench_passes.py --only lighting > $O/passes.txt 2>&1; cat $O/passes.txt
