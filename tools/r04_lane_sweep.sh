#!/bin/bash
# round 4: parity of the three path kernels, then the path kernel alone by lanes per ray and rays in the launch
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r04_parity.log 2>&1 || { tail -30 gpurun_out/r04_parity.log; exit 1; }
tail -3 gpurun_out/r04_parity.log
for L in 1 2 4; do
    RVB_PATH_LANES=$L timeout -k 10 300 python tools/rays_sweep.py 100000 200000 400000 800000 > gpurun_out/r04_sweep_lanes$L.txt 2>&1
    cat gpurun_out/r04_sweep_lanes$L.txt
done
for L in 1 2; do
    for G in 2 4; do
        RVB_PATH_LANES=$L SWEEP_GROUP=$G timeout -k 10 300 python tools/rays_sweep.py 100000 > gpurun_out/r04_sweep_lanes${L}_group$G.txt 2>&1
        cat gpurun_out/r04_sweep_lanes${L}_group$G.txt
    done
done
