#!/bin/bash
# round 4: the builder's leaf policy (full leaves of up to four triangles against the SAH's own decision) with the final kernels, in the pipeline
cd "$(dirname "$0")/.."
out=gpurun_out/r04d_leaf_policy_n1.txt
: > $out
for rep in 1 2; do
    echo "pipeline, full leaves (shipped): $(python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
    echo "pipeline, RVB_LEAF_POLICY=sah: $(RVB_LEAF_POLICY=sah python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
done
cat $out
