#!/bin/bash
# rasteriser: parity tests, then the producer passes' times (sparse and dense atrium)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_raster.py tests/test_golden.py tests/test_textures.py -x -q -m gpu > gpurun_out/r3_raster_tests.log 2>&1 || { tail -30 gpurun_out/r3_raster_tests.log; exit 1; }
tail -2 gpurun_out/r3_raster_tests.log
timeout -k 10 300 python tools/bench_passes.py --only raster --iters 20 2>/dev/null | grep -i raster
