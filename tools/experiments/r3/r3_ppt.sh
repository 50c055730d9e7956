#!/bin/bash
for w in 4k_deferred_gi_random 4k_deferred_gi; do for p in 4 2 1; do
SAH_FORCE_PPT=$p python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w ppt $p', d['ms_per_step'])"
done; done
