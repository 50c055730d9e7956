#!/bin/bash
# The randomised HIP-vs-oracle sweeps at larger case counts than the refresh runs them (about ten minutes on one MI355X box).
# Logs under gpurun_out/stress/, copied by hand into profiles/rN_stress_*_long.txt.
set -u
O=gpurun_out/stress
rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 tools/stress_rt.py --cases 160 > $O/stress_rt.txt 2>&1; echo "rt: $(tail -1 $O/stress_rt.txt)"
timeout -k 10 600 python3 tools/stress_raster.py --cases 120 > $O/stress_raster.txt 2>&1; echo "raster: $(tail -1 $O/stress_raster.txt)"
timeout -k 10 600 python3 tools/stress_raster.py --textured --cases 120 > $O/stress_raster_textured.txt 2>&1; echo "raster textured: $(tail -1 $O/stress_raster_textured.txt)"
timeout -k 10 600 python3 tools/stress_parity.py --seeds 40 --big > $O/stress_parity.txt 2>&1; echo "parity: $(tail -1 $O/stress_parity.txt)"
timeout -k 10 600 python3 tools/stress_post.py --cases 240 > $O/stress_post.txt 2>&1; echo "post: $(tail -1 $O/stress_post.txt | cut -c1-200)"
