#!/bin/bash
# round 4, after the pair path kernel's node step was rewritten (pushes from counts, signed keys): the randomized parity campaign again,
# two lanes per ray forced (the rewritten kernel; small cases default to the four-lane kernel) and the default choice
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
make -C oracle > /dev/null 2>&1
RVB_PATH_LANES=2 timeout -k 10 900 python tools/fuzz_parity.py 120 gpu > gpurun_out/r04c_fuzz_parity_pairs_gpu.log 2>&1; echo "pairs rc $?"; tail -1 gpurun_out/r04c_fuzz_parity_pairs_gpu.log
RVB_PATH_LANES=2 FUZZ_REFLECTION_MULTIPLE=32 timeout -k 10 900 python tools/fuzz_parity.py 60 gpu > gpurun_out/r04c_fuzz_parity_pairs_keyruns_gpu.log 2>&1; echo "pairs key runs rc $?"; tail -1 gpurun_out/r04c_fuzz_parity_pairs_keyruns_gpu.log
timeout -k 10 900 python tools/fuzz_parity.py 60 gpu > gpurun_out/r04c_fuzz_parity_default_gpu.log 2>&1; echo "default rc $?"; tail -1 gpurun_out/r04c_fuzz_parity_default_gpu.log
