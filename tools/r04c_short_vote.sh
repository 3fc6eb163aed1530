#!/bin/bash
# round 4: the pair path kernel's vote in its short form (RVB_PAIR_SHORT_VOTE: one ballot when the lanes at a node are half of the live lanes or more)
cd "$(dirname "$0")/.."
V=parallel-reverb-raytracer_amd/_variants
out=gpurun_out/${OUT:-r04c_short_vote_n1}.txt; VARIANTS=${VARIANTS:-"cur sv"}
: > $out
for v in $VARIANTS; do
    if ! RVB_LIB=$PWD/$V/lib_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seeded or golden" > gpurun_out/ab_sv_$v.parity.log 2>&1; then echo "$v PARITY-FAIL" >> $out; fi
    echo "alone, $v: $(RVB_PATH_LANES=2 RVB_LIB=$PWD/$V/lib_$v.so python tools/rays_sweep.py 100000 196608 800000 2>&1 | grep -v amdgpu | cut -c1-60 | tr '\n' ';')" >> $out
done
for rep in 1 2 3; do
    for v in $VARIANTS; do
        echo "pipeline, $v: $(RVB_LIB=$PWD/$V/lib_$v.so python bench.py --steps 160 --warmup 12 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" >> $out
    done
done
cat $out
