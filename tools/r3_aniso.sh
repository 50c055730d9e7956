#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_textures.py tests/test_raster.py tests/test_golden.py tests/test_host_facade_gpu.py tests/test_rt.py tests/test_lpv_inject.py -x -q -m gpu > gpurun_out/r3_aniso_tests.log 2>&1 || { tail -30 gpurun_out/r3_aniso_tests.log; exit 1; }
tail -2 gpurun_out/r3_aniso_tests.log
timeout -k 10 600 python tools/stress_raster.py --textured --cases 40 > gpurun_out/r3_stress_raster_textured.log 2>&1 || { tail -20 gpurun_out/r3_stress_raster_textured.log; exit 1; }
tail -2 gpurun_out/r3_stress_raster_textured.log
python tools/bench_passes.py --only raster 2>&1 | grep -v "^/opt"
