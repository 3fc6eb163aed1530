// Developer tool: what HBM-to-HBM rate do different launch shapes of a 16-byte-per-lane streaming kernel reach on this
// GPU?  (Sets the target for attenuate_kernel.)  hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT, bool SPAN>
__global__ __launch_bounds__(256) void copy_kernel(const f4 * __restrict__ in, f4 * __restrict__ out, size_t n)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    if (SPAN) {
        // each block owns a contiguous span; inside it the block sweeps 4 KiB * UNROLL per iteration
        const size_t per_block = (n + gridDim.x - 1) / gridDim.x;
        const size_t begin = (size_t) blockIdx.x * per_block, end = begin + per_block < n ? begin + per_block : n;
        for (size_t c0 = begin + threadIdx.x; c0 < end; c0 += (size_t) blockDim.x * UNROLL) {
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { size_t c = c0 + (size_t) u * blockDim.x; if (c < end) v[u] = NT ? __builtin_nontemporal_load(in + c) : in[c]; }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { size_t c = c0 + (size_t) u * blockDim.x; if (c < end) { if (NT) __builtin_nontemporal_store(v[u], out + c); else out[c] = v[u]; } }
        }
    } else {
        for (size_t c0 = (size_t) blockIdx.x * blockDim.x + threadIdx.x; c0 < n; c0 += stride * UNROLL) {
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { size_t c = c0 + (size_t) u * stride; if (c < n) v[u] = NT ? __builtin_nontemporal_load(in + c) : in[c]; }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { size_t c = c0 + (size_t) u * stride; if (c < n) { if (NT) __builtin_nontemporal_store(v[u], out + c); else out[c] = v[u]; } }
        }
    }
}

template <int UNROLL, bool NT, bool SPAN>
void run(const char * name, const f4 * in, f4 * out, size_t n, unsigned blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((copy_kernel<UNROLL, NT, SPAN>), dim3(blocks), dim3(256), 0, 0, in, out, n);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r && ms < best) best = ms;
    }
    printf("%-28s blocks %6u  %.3f ms  %.2f TB/s\n", name, blocks, best, 2.0 * n * 16 / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t n = 12800000ull * 4;            // 16-byte chunks of 12.8 M impulses = 819 MB
    f4 * in, * out;
    hipMalloc(&in, n * 16); hipMalloc(&out, n * 16);
    hipMemset(in, 1, n * 16); hipMemset(out, 0, n * 16);
    hipDeviceSynchronize();
    for (unsigned blocks : {2048u, 4096u, 8192u, 16384u, 200000u}) {
        run<1, false, false>("stride u1 plain", in, out, n, blocks);
        run<4, false, false>("stride u4 plain", in, out, n, blocks);
        run<4, true, false>("stride u4 nt", in, out, n, blocks);
        run<8, true, false>("stride u8 nt", in, out, n, blocks);
        run<4, false, true>("span u4 plain", in, out, n, blocks);
        run<4, true, true>("span u4 nt", in, out, n, blocks);
        run<8, true, true>("span u8 nt", in, out, n, blocks);
    }
    hipDeviceSynchronize();
    return 0;
}
