// Developer tool: is 1.0f / x (hipcc's correctly rounded division sequence: v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas,
// v_div_fixup) reproduced bit for bit by v_rcp_f32 + FMA refinement without the scaling and fix-up steps, for EVERY float x
// with 2^-17 <= |x| <= 2^64 (the determinants the Möller–Trumbore test divides by are >= 1e-4 in magnitude)?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o tools/_bin/rcp_probe tools/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float rcp_a(float x)      // the division sequence minus scaling / fix-up: 7 instructions
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e0 = fmaf(-x, r0, 1.0f);
    const float r1 = fmaf(e0, r0, r0);
    const float e1 = fmaf(-x, r1, 1.0f);
    const float q1 = fmaf(e1, r1, r1);
    const float e2 = fmaf(-x, q1, 1.0f);
    return fmaf(e2, r1, q1);
}
__device__ __forceinline__ float rcp_b(float x)      // 5 instructions
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e0 = fmaf(-x, r0, 1.0f);
    const float r1 = fmaf(e0, r0, r0);
    const float e1 = fmaf(-x, r1, 1.0f);
    return fmaf(e1, r1, r1);
}
__device__ __forceinline__ float rcp_c(float x)      // 3 instructions
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e0 = fmaf(-x, r0, 1.0f);
    return fmaf(e0, r0, r0);
}

__global__ void probe(uint32_t lo_bits, uint32_t hi_bits, unsigned long long * bad)
{
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    unsigned long long a = 0, b = 0, c = 0;
    for (uint64_t i = lo_bits + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= hi_bits; i += stride) {
        for (int sign = 0; sign < 2; ++sign) {
            const float x = __uint_as_float((uint32_t) i | (sign ? 0x80000000u : 0u));
            const float want = 1.0f / x;
            a += __float_as_uint(rcp_a(x)) != __float_as_uint(want);
            b += __float_as_uint(rcp_b(x)) != __float_as_uint(want);
            c += __float_as_uint(rcp_c(x)) != __float_as_uint(want);
        }
    }
    if (a) atomicAdd(bad + 0, a);
    if (b) atomicAdd(bad + 1, b);
    if (c) atomicAdd(bad + 2, c);
}

int main()
{
    unsigned long long * bad, host[3] = {0, 0, 0};
    hipMalloc(&bad, sizeof(host));
    hipMemset(bad, 0, sizeof(host));
    const float lo = 0x1p-17f, hi = 0x1p64f;
    uint32_t lo_bits, hi_bits;
    memcpy(&lo_bits, &lo, 4); memcpy(&hi_bits, &hi, 4);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, lo_bits, hi_bits, bad);
    hipMemcpy(host, bad, sizeof(host), hipMemcpyDeviceToHost);
    printf("floats tested: %llu (both signs); mismatches vs 1.0f/x: 7-instr %llu, 5-instr %llu, 3-instr %llu\n",
           2ull * (hi_bits - lo_bits + 1ull), host[0], host[1], host[2]);
    return 0;
}
