#!/bin/bash
# round 4: the pair path kernel's node step rewritten for issue cost (pushes from the lanes' counts, signed keys, the culling distance as
# state, v_bitop3 keys: RVB_PAIR_PUSH_COUNTS) against the hit-mask form, each under register caps of 6 and 7 waves per SIMD (80 / 72
# VGPRs; RVB_PAIR_WAVES, RVB_SHADOW_PAIR_WAVES) — same C-ABI, the pipeline of bench.py
cd "$(dirname "$0")/.."
V=parallel-reverb-raytracer_amd/_variants
out=gpurun_out/r04_push_counts_n1.txt
: > $out
variants="base new w6 w7 w6s6 w7s7 basew6 basew7"
for v in $variants; do
    if ! RVB_LIB=$PWD/$V/lib_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seeded or golden" > gpurun_out/ab_push_$v.parity.log 2>&1; then
        echo "$v PARITY-FAIL" >> $out
    fi
done
for rep in 1 2 3; do
    for v in $variants; do
        echo "pipeline, $v: $(RVB_LIB=$PWD/$V/lib_$v.so python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
    done
done
cat $out
