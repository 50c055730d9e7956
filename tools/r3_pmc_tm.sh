#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_pmc_tm; mkdir -p $O
PMC_SCRIPT=tools/bench_passes.py PMC_ARGS="--only tonemap --iters 5" PMC_KERNEL=k_tonemap bash tools/pmc_collect.sh $O/pmc > $O/pmc_tonemap.txt 2>&1
cat $O/pmc_tonemap.txt | grep -v "^pmc group" | head -90
