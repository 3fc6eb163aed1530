// Developer tool: what a counting-sort style record grouping costs on the GPU — 12.8 M records, 2^16 buckets with a skewed
// distribution: (a) no-return atomic count, (b) single-workgroup scan, (c) fill with returning atomics + scattered 4-byte
// stores, (d) fill from precomputed ranks (no atomics).   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/group_probe tools/group_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void count_kernel(const uint32_t * keys, uint32_t n, uint32_t * count)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) atomicAdd(count + keys[i], 1u);
}
__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t * count, uint32_t * start, uint32_t nb)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023) / 1024, lo = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t i = lo; i < lo + per && i < nb; ++i) s += count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < lo + per && i < nb; ++i) { start[i] = run; run += count[i]; }
}
__global__ __launch_bounds__(64) void fill_atomic_kernel(const uint32_t * keys, uint32_t n, uint32_t * cursor, uint32_t * order)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) { const uint32_t pos = atomicAdd(cursor + keys[i], 1u); order[pos] = i; }
}
__global__ __launch_bounds__(64) void fill_rank_kernel(const uint32_t * keys, const uint32_t * rank, const uint32_t * start, uint32_t n, uint32_t * order)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) order[start[keys[i]] + rank[i]] = i;
}

int main()
{
    const uint32_t n = 12800000, nb = 65536;
    std::vector<uint32_t> keys(n), rank(n), cnt(nb, 0);
    uint64_t s = 12345;
    for (uint32_t i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        double u = (double) (s >> 11) / 9007199254740992.0;
        uint32_t k = (uint32_t) (u * u * nb);          // skewed: low buckets are hit more often
        keys[i] = k < nb ? k : nb - 1;
        rank[i] = cnt[keys[i]]++;
    }
    uint32_t *d_keys, *d_rank, *d_count, *d_start, *d_cursor, *d_order;
    CHECK(hipMalloc(&d_keys, n * 4)); CHECK(hipMalloc(&d_rank, n * 4)); CHECK(hipMalloc(&d_count, nb * 4)); CHECK(hipMalloc(&d_start, nb * 4));
    CHECK(hipMalloc(&d_cursor, nb * 4)); CHECK(hipMalloc(&d_order, n * 4));
    CHECK(hipMemcpy(d_keys, keys.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rank, rank.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e[5];
    for (auto & x : e) CHECK(hipEventCreate(&x));
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemset(d_count, 0, nb * 4));
        CHECK(hipEventRecord(e[0]));
        hipLaunchKernelGGL(count_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, d_keys, n, d_count);
        CHECK(hipEventRecord(e[1]));
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, 0, d_count, d_start, nb);
        CHECK(hipMemcpyAsync(d_cursor, d_start, nb * 4, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(e[2]));
        hipLaunchKernelGGL(fill_atomic_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, d_keys, n, d_cursor, d_order);
        CHECK(hipEventRecord(e[3]));
        hipLaunchKernelGGL(fill_rank_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, d_keys, d_rank, d_start, n, d_order);
        CHECK(hipEventRecord(e[4]));
        CHECK(hipDeviceSynchronize());
        float t[4];
        for (int i = 0; i < 4; ++i) CHECK(hipEventElapsedTime(&t[i], e[i], e[i + 1]));
        printf("count %.3f ms | scan+copy %.3f ms | fill (returning atomics) %.3f ms | fill (ranks) %.3f ms\n", t[0], t[1], t[2], t[3]);
    }
    return 0;
}
