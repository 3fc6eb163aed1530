#!/bin/bash
# round 4: the bench pipeline with the path kernel forced to 1 / 2 lanes per ray, 4 contexts in groups of 2 and 8 in groups of 4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r04_pipeline_lanes_n1.txt
: > $out
for cfg in "2 4" "1 4" "2 8" "1 8"; do
    set -- $cfg
    echo "lanes $1 contexts $2" >> $out
    RVB_PATH_LANES=$1 timeout -k 10 300 python bench.py --steps 100 --warmup 8 --contexts $2 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep "timed region" >> $out
done
cat $out
