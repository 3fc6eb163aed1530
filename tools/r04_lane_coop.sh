#!/bin/bash
# round 4: the one-lane path kernel with the cooperative node fetch — parity, then the path kernel alone by rays in the launch
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "one_lane" > gpurun_out/r04_parity_coop.log 2>&1 || { tail -30 gpurun_out/r04_parity_coop.log; exit 1; }
tail -2 gpurun_out/r04_parity_coop.log
for L in 1 2; do
    RVB_PATH_LANES=$L timeout -k 10 300 python tools/rays_sweep.py 100000 200000 400000 800000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_sweep_coop_lanes$L.txt
done
for G in 2 4; do
    RVB_PATH_LANES=1 SWEEP_GROUP=$G timeout -k 10 300 python tools/rays_sweep.py 100000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_sweep_coop_lanes1_group$G.txt
done
