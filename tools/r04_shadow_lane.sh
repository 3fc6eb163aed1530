#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "quad_shadow" 2>&1 | tail -3
out=gpurun_out/r04_shadow_lanes_n1.txt
: > $out
for L in 2 1 4; do
    RVB_SHADOW_LANES=$L timeout -k 10 300 python tools/rays_sweep.py 100000 400000 2>&1 | grep -v amdgpu.ids | sed "s/^/shadow lanes $L | /" >> $out
done
for L in 2 1; do
    RVB_SHADOW_LANES=$L timeout -k 10 300 python bench.py --steps 200 --warmup 8 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep "timed region" | sed "s/^/shadow lanes $L pipeline: /" >> $out
done
cat $out
