#!/bin/bash
# Developer tool: builds parallel-reverb-raytracer_amd/_variants/lib_<name>.so = the shipped library with
# trace_kernels.hip recompiled under extra flags (A/B experiments over the same C-ABI, see tools/ab_bench.sh).
set -e
name=$1; shift
cd "$(dirname "$0")/../parallel-reverb-raytracer_amd"
make -j4 librvb_hip.so > /dev/null
mkdir -p _variants _build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function "$@" \
    -c csrc/trace_kernels.hip -o _build/variants/trace_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/lib_$name.so _build/capi.o _build/bvh_build.o _build/variants/trace_$name.o _build/stream_kernels.o _build/radix_sort.o _build/multi.o _build/pipeline.o -ldl -lpthread
echo built _variants/lib_$name.so
