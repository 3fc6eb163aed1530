#!/bin/bash
# round 4: a seventh wave per SIMD once more, with the cold ray state in LDS (RVB_PAIR_COLD=1, 72 VGPRs) and LDS room for it (no key runs: 5.4 KB per workgroup)
cd "$(dirname "$0")/.."
V=parallel-reverb-raytracer_amd/_variants
out=gpurun_out/r04d_seven_waves_n1.txt
: > $out
if ! RVB_KEY_RUNS=0 RVB_LIB=$PWD/$V/lib_cold7.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seeded or golden" > gpurun_out/ab_cold7.parity.log 2>&1; then echo "cold7 PARITY-FAIL" >> $out; fi
for rep in 1 2 3; do
    for cfg in "cur 1" "cur 0" "cold7 0" "cold7 1"; do
        set -- $cfg
        echo "pipeline, $1, key runs $2: $(RVB_KEY_RUNS=$2 RVB_LIB=$PWD/$V/lib_$1.so python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
    done
done
cat $out
