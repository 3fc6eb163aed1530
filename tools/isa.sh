#!/bin/bash
# Developer tool: gfx950 ISA of trace_kernels.hip (extra flags pass through) -> /tmp/isa/<name>.s, plus per-kernel files.
#   tools/isa.sh <name> [-Dflags...]
set -e
name=$1; shift
mkdir -p /tmp/isa
cd "$(dirname "$0")/../parallel-reverb-raytracer_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -fno-slp-vectorize --cuda-device-only "$@" -S -o /tmp/isa/$name.s csrc/trace_kernels.hip 2>&1 | grep -E "error" -A3 || true
for k in path image shadow; do
    awk "/^_ZN12_GLOBAL__N_1[0-9]+${k}_kernel(ILb1E[A-Za-z0-9]*Ev)?9?.*TraceArgs:/,/s_endpgm/" /tmp/isa/$name.s > /tmp/isa/${name}_$k.s
done
grep -E "vgpr_count|Spill|Reload" /tmp/isa/$name.s | sort | uniq -c
