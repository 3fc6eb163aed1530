#!/bin/bash
# round 4: the record grouping's key width in the PIPELINE (it was chosen with one IR alone on the GPU): 16 bits = two onesweep passes, 8 = one
cd "$(dirname "$0")/.."
out=gpurun_out/r04d_sort_bits_n1.txt
: > $out
for rep in 1 2; do
    for bits in 16 12 8 6; do
        echo "pipeline, grouping key bits $bits: $(RVB_SHADOW_SORT_BITS=$bits python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
    done
    echo "pipeline, no grouping: $(RVB_SHADOW_SORT=0 python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
done
cat $out
