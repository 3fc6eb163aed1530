#!/bin/bash
# round 4: one copy of the scene for all the contexts of a GPU (rvb_share_scene) against a copy per context — the pipeline of bench.py
cd "$(dirname "$0")/.."
out=gpurun_out/r04c_share_scene_n1.txt
: > $out
for rep in 1 2 3; do
    for flag in "" "--own-scenes"; do
        echo "pipeline, 4 contexts ${flag:-shared scene}: $(python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline $flag 2>&1 >/dev/null | grep 'timed region')" >> $out
    done
done
cat $out
