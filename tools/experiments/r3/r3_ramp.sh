#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_ramp; mkdir -p $O
for r in 0 50 200 1000; do
  for rep in 1 2; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --ramp-ms $r 2>>$O/err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ramp $r', d['ms_per_step'], d['roofline']['kernel_ms_mean'])"
  done
done
python3 bench.py --no-cpu-baseline 2>>$O/err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('200 steps', d['ms_per_step'], d['roofline']['kernel_ms_mean'])"
python3 bench.py --no-cpu-baseline --steps 2000 --warmup 50 2>>$O/err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('2000 steps', d['ms_per_step'], d['roofline']['kernel_ms_mean'])"
