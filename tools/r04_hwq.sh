#!/bin/bash
# round 4: hardware queues per process (GPU_MAX_HW_QUEUES, default 4) — the pipeline has 13+ HIP streams; streams that share a hardware queue serialize
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r04_hw_queues_n1.txt
: > $out
for q in 4 8 16 24 2; do
    for drv in "--native" ""; do
        GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 200 --warmup 8 $drv --no-extras --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
        echo "GPU_MAX_HW_QUEUES=$q driver '$drv': $(grep 'timed region' /tmp/b.err)" >> $out
    done
done
cat $out
