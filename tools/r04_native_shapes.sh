#!/bin/bash
# round 4: shapes of the native pipeline (contexts / traces per group launch)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/${SHAPES_TAG:-r04_native_pipeline_shapes_n1}.txt
: > $out
for cfg in "4 2" "8 4" "6 3" "8 2" "6 2" "4 1" "2 1" "4 4" "8 4"; do
    set -- $cfg
    RVB_PIPELINE_GROUP=$2 timeout -k 10 300 python bench.py --steps 200 --warmup 16 --native --contexts $1 --no-extras --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
    echo "contexts $1 group $2: $(grep 'timed region' /tmp/b.err)" >> $out
done
cat $out
