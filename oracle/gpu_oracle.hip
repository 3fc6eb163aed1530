// TEST INFRASTRUCTURE ONLY.  The kernel part of oracle/rvb_oracle.c — the plain-C restatement of reference kernel `raytrace`
// (rayverb/kernel.cpp:304-503) — compiled for the GPU as it stands: one thread per ray, brute force over all triangles for every
// query, no acceleration structure, no shared code with the product.  Its only purpose is to check WHOLE full-size runs of the
// product (every ray of BASELINE configs C2 / C3 / C4, 10^12 triangle tests) where the CPU build of the same source can only
// afford a sample of rays.  tests/ cross-check it against the CPU build on that sample first.
//
// Arithmetic: -ffp-contract=off, correctly rounded divide / sqrt, one IEEE operation per operator like the CPU build.  pow() is
// the device library's binary64 pow, rounded to float: the same correctly rounded float as glibc's except where the two
// binary64 results straddle a float rounding boundary (~1e-8 per value), so volumes are compared with a one-ulp allowance and
// the number of such values is reported; positions, times and image-source indices must match bit for bit.
#include <hip/hip_runtime.h>

#define RVBO_FN __host__ __device__
#define RVBO_KERNEL_PART_ONLY
#include "rvb_oracle.c"

#include <cstdio>

__global__ __launch_bounds__(64) void gpu_oracle_kernel(const float * directions, uint64_t first, uint64_t count, uint64_t total,
                                                        const RvboTriangle * triangles, uint64_t ntriangles, const float * vertices,
                                                        const RvboSurface * surfaces, v3 mic, v3 source, uint64_t nreflections,
                                                        const float * air, RvboImpulse * impulses, RvboImpulse * image_source,
                                                        uint64_t * image_source_index)
{
    const uint64_t i = first + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + count || i >= total)
        return;
    float air8[8];
    for (int b = 0; b < 8; ++b) air8[b] = air[b];
    raytrace_one(i, directions, mic, triangles, ntriangles, vertices, source, surfaces, impulses, image_source, image_source_index,
                 nreflections, air8);
}

#define GO(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "gpu_oracle: %s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

// Same contract as rvbo_raytrace (rvb_oracle.h): host arrays in, host arrays out (outputs zero-filled first, rayverb.cpp:600-616).
extern "C" int rvbo_gpu_raytrace(const float * directions, uint64_t nrays, const RvboTriangle * triangles, uint64_t ntriangles,
                                 const float * vertices, uint64_t nvertices, const RvboSurface * surfaces, uint64_t nsurfaces,
                                 const float mic[3], const float source[3], uint64_t nreflections, const float air[8],
                                 RvboImpulse * impulses, RvboImpulse * image_source, uint64_t * image_source_index, int device)
{
    GO(hipSetDevice(device));
    float * d_dirs = nullptr, * d_verts = nullptr, * d_air = nullptr;
    RvboTriangle * d_tris = nullptr;
    RvboSurface * d_surf = nullptr;
    RvboImpulse * d_imp = nullptr, * d_img = nullptr;
    uint64_t * d_idx = nullptr;
    const size_t imp_bytes = sizeof(RvboImpulse) * nrays * nreflections, img_bytes = sizeof(RvboImpulse) * nrays * RVBO_NUM_IMAGE_SOURCE,
                 idx_bytes = sizeof(uint64_t) * nrays * RVBO_NUM_IMAGE_SOURCE;
    GO(hipMalloc(&d_dirs, nrays * 16 + 16)); GO(hipMalloc(&d_verts, nvertices * 16 + 16)); GO(hipMalloc(&d_air, 32));
    GO(hipMalloc(&d_tris, ntriangles * sizeof(RvboTriangle) + 16)); GO(hipMalloc(&d_surf, nsurfaces * sizeof(RvboSurface) + 16));
    GO(hipMalloc(&d_imp, imp_bytes + 16)); GO(hipMalloc(&d_img, img_bytes + 16)); GO(hipMalloc(&d_idx, idx_bytes + 16));
    GO(hipMemcpy(d_dirs, directions, nrays * 16, hipMemcpyHostToDevice));
    GO(hipMemcpy(d_verts, vertices, nvertices * 16, hipMemcpyHostToDevice));
    GO(hipMemcpy(d_air, air, 32, hipMemcpyHostToDevice));
    GO(hipMemcpy(d_tris, triangles, ntriangles * sizeof(RvboTriangle), hipMemcpyHostToDevice));
    GO(hipMemcpy(d_surf, surfaces, nsurfaces * sizeof(RvboSurface), hipMemcpyHostToDevice));
    GO(hipMemset(d_imp, 0, imp_bytes)); GO(hipMemset(d_img, 0, img_bytes)); GO(hipMemset(d_idx, 0, idx_bytes));
    const v3 m = v3_make(mic[0], mic[1], mic[2]), s = v3_make(source[0], source[1], source[2]);
    // launches of 16 384 rays: each finishes in about a second even at 263 k triangles x 256 bounces
    const uint64_t per_launch = 16384;
    for (uint64_t first = 0; first < nrays; first += per_launch) {
        const uint64_t count = nrays - first < per_launch ? nrays - first : per_launch;
        hipLaunchKernelGGL(gpu_oracle_kernel, dim3((unsigned) ((count + 63) / 64)), dim3(64), 0, 0, d_dirs, first, count, nrays, d_tris, ntriangles,
                           d_verts, d_surf, m, s, nreflections, d_air, d_imp, d_img, d_idx);
        GO(hipGetLastError());
        GO(hipDeviceSynchronize());
    }
    GO(hipMemcpy(impulses, d_imp, imp_bytes, hipMemcpyDeviceToHost));
    GO(hipMemcpy(image_source, d_img, img_bytes, hipMemcpyDeviceToHost));
    GO(hipMemcpy(image_source_index, d_idx, idx_bytes, hipMemcpyDeviceToHost));
    for (void * p : {(void *) d_dirs, (void *) d_verts, (void *) d_air, (void *) d_tris, (void *) d_surf, (void *) d_imp, (void *) d_img, (void *) d_idx})
        (void) hipFree(p);
    return 0;
}
