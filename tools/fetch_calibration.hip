// Developer tool: what do FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report per byte for the access patterns of this library's
// kernels?  MI355X_MICROARCH.md: FETCH_SIZE tallies a wide coalesced streaming read at HALF its bytes (128-byte requests counted as
// 64) and says "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Four kernels,
// each moving a known number of bytes of a 1 GiB array of 64-byte records (far beyond the 256 MiB Infinity Cache):
//   stream_read      16 bytes per lane, consecutive (the streaming kernels)
//   gather64_read    one 64-byte record per lane PAIR at a pseudo-random index, each lane 2 x 16 bytes of it (ordered_sum_kernel)
//   gather64_rw      the same record read (2 lanes x 32 bytes) and rewritten in place (shadow_pair_kernel)
//   scatter64_write  one 64-byte record per lane pair written at a pseudo-random index (path kernels' record stores, per 2 lanes)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fetch_calibration tools/fetch_calibration.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/_bin/fetch_calibration     (and WRITE_SIZE in a pass of its own)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void stream_read(const f4 * __restrict__ in, float * __restrict__ sink, size_t n16)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) acc += in[i];
    if (acc.x == 1.2345f) sink[0] = acc.y + acc.z + acc.w;
}
// records: n of them, `count` gathers
__global__ __launch_bounds__(64) void gather64_read(const f4 * __restrict__ in, float * __restrict__ sink, uint32_t nrec, uint32_t count)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, g = t >> 1, half = t & 1u;
    if (g >= count) return;
    const uint32_t rec = mix(g) % nrec;
    const f4 a = in[4u * (size_t) rec + half], b = in[4u * (size_t) rec + 2];
    if (a.x + b.x == 1.2345f) sink[0] = a.y;
}
__global__ __launch_bounds__(64) void gather64_rw(f4 * __restrict__ data, uint32_t nrec, uint32_t count)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, g = t >> 1, half = t & 1u;
    if (g >= count) return;
    // a permutation of the records (odd multiplier modulo a power of two): every record read and written exactly once
    const uint32_t rec = (g * 2654435761u) & (nrec - 1u);
    f4 a = __builtin_nontemporal_load(data + 4u * (size_t) rec + half), b = __builtin_nontemporal_load(data + 4u * (size_t) rec + half + 2);
    a += 1.0f; b += 1.0f;
    __builtin_nontemporal_store(a, data + 4u * (size_t) rec + half);
    __builtin_nontemporal_store(b, data + 4u * (size_t) rec + half + 2);
}
__global__ __launch_bounds__(64) void scatter64_write(f4 * __restrict__ data, uint32_t nrec, uint32_t count)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, g = t >> 1, half = t & 1u;
    if (g >= count) return;
    const uint32_t rec = (g * 2654435761u) & (nrec - 1u);
    const f4 v = {(float) g, 1.0f, 2.0f, 3.0f};
    __builtin_nontemporal_store(v, data + 4u * (size_t) rec + half);
    __builtin_nontemporal_store(v, data + 4u * (size_t) rec + half + 2);
}

int main()
{
    const uint32_t nrec = 1u << 24;                       // 16 Mi records x 64 B = 1 GiB
    const uint32_t count = 8u << 20;                      // 8 Mi gathers = 512 MiB of records
    f4 * data; float * sink;
    if (hipMalloc(&data, (size_t) nrec * 64) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("no GPU memory\n"); return 2; }
    hipMemset(data, 0, (size_t) nrec * 64);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_read, dim3((unsigned) (((size_t) nrec * 4 + 255) / 256)), dim3(256), 0, 0, data, sink, (size_t) nrec * 4);
        hipLaunchKernelGGL(gather64_read, dim3(count * 2 / 64), dim3(64), 0, 0, data, sink, nrec, count);
        hipLaunchKernelGGL(gather64_rw, dim3(nrec * 2 / 64), dim3(64), 0, 0, data, nrec, nrec);
        hipLaunchKernelGGL(scatter64_write, dim3(nrec * 2 / 64), dim3(64), 0, 0, data, nrec, nrec);
        hipDeviceSynchronize();
    }
    printf("bytes moved per launch: stream_read %.1f MiB read; gather64_read %.1f MiB of 64-byte records read; gather64_rw %.1f MiB read + %.1f MiB written; scatter64_write %.1f MiB written\n",
           nrec * 64.0 / 1048576, count * 64.0 / 1048576, nrec * 64.0 / 1048576, nrec * 64.0 / 1048576, nrec * 64.0 / 1048576);
    return 0;
}
