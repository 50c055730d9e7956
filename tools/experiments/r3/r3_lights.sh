#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_lighting_ext_gpu.py tests/test_lighting_gpu.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
for w in 1080p_64_lights 4k_256_lights 8k_1024_lights_gi; do
python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['roofline']['kernel_ms_mean'])"
done
