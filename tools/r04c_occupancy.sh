#!/bin/bash
# round 4: does a seventh wave per SIMD pay?  The pair path kernel at 80 / 72 / 64 VGPRs (RVB_PAIR_WAVES 6 / 7 / 8; 72 spills ten registers), with
# and without the key runs in LDS (RVB_KEY_RUNS=0: 4.1 instead of 6.1 KB per workgroup, so that LDS allows more than six waves too)
cd "$(dirname "$0")/.."
V=parallel-reverb-raytracer_amd/_variants
out=gpurun_out/r04c_occupancy_n1.txt
: > $out
for rep in 1 2; do
    for cfg in "w6 1" "w6 0" "w7 1" "w7 0" "w7s7 0" "w8 0"; do
        set -- $cfg
        echo "pipeline, $1, key runs $2: $(RVB_KEY_RUNS=$2 RVB_LIB=$PWD/$V/lib_$1.so python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
    done
done
cat $out
