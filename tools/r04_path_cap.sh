#!/bin/bash
# round 4: the two-lane path kernel capped to 5 / 4 waves per SIMD by its LDS request (8 / 10 KiB per workgroup), pipeline shapes beside it
cd "$(dirname "$0")/.."
out=gpurun_out/r04_path_wave_cap_n1.txt
: > $out
for cfg in "0 4 2" "8192 4 2" "10240 4 2" "8192 4 1" "10240 4 1" "8192 6 2" "10240 6 2" "10240 8 2" "8192 6 3" "0 4 2"; do
    set -- $cfg
    echo "path LDS floor $1 B, contexts $2 group $3: $(RVB_PATH_LDS_BYTES=$1 RVB_PIPELINE_GROUP=$3 RVB_PATH_LANES=2 python bench.py --steps 160 --warmup 12 --contexts $2 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
done
cat $out
