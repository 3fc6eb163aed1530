#!/bin/bash
# round 4: randomized parity campaign (GPU brute-force oracle as the checker) with the one-lane path kernel, whole key runs, and the one-lane shadow kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
make -C oracle > /dev/null 2>&1
RVB_PATH_LANES=1 timeout -k 10 900 python tools/fuzz_parity.py 60 gpu > gpurun_out/r04_fuzz_parity_lanes1_gpu.log 2>&1; echo "lanes1 rc $?"; tail -1 gpurun_out/r04_fuzz_parity_lanes1_gpu.log
RVB_PATH_LANES=1 FUZZ_REFLECTION_MULTIPLE=32 timeout -k 10 900 python tools/fuzz_parity.py 45 gpu > gpurun_out/r04_fuzz_parity_lanes1_keyruns_gpu.log 2>&1; echo "lanes1 key runs rc $?"; tail -1 gpurun_out/r04_fuzz_parity_lanes1_keyruns_gpu.log
RVB_SHADOW_LANES=1 timeout -k 10 900 python tools/fuzz_parity.py 30 gpu > gpurun_out/r04_fuzz_parity_shadow_lanes1_gpu.log 2>&1; echo "shadow lanes1 rc $?"; tail -1 gpurun_out/r04_fuzz_parity_shadow_lanes1_gpu.log
timeout -k 10 900 python tools/fuzz_parity.py 45 gpu > gpurun_out/r04_fuzz_parity_default_gpu.log 2>&1; echo "default rc $?"; tail -1 gpurun_out/r04_fuzz_parity_default_gpu.log
