#!/bin/bash
# round 4: the shadow kernel's grid (single-wave workgroups per CU; the records beyond are grid-strided) in the PIPELINE
cd "$(dirname "$0")/.."
out=gpurun_out/r04d_shadow_grid_n1.txt
: > $out
for rep in 1 2; do
    for n in 256 24 48 96 512 1600; do
        echo "pipeline, shadow workgroups per CU $n: $(RVB_SHADOW_WG_PER_CU=$n python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
    done
done
cat $out
