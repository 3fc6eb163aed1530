// trace_kernels.hip — the per-ray trace of reference rayverb/kernel.cpp:304-503 (kernel
// `raytrace`), re-organised for CDNA4 as three kernels over one 4-wide BVH:
//
//   path_kernel    one lane per ray: the inherently sequential chain closest hit -> reflect
//                  (kernel.cpp:359-375, :459-461, :478, :492-501).  Latency-bound; per bounce it
//                  leaves a 64-byte work record in the ray's Impulse slot.
//   image_kernel   one lane per (ray, bounce < 9): image-source validation (kernel.cpp:379-457).
//                  The reference does this inside the ray's loop; its inputs are only the
//                  triangles the ray hit so far, so it parallelises 9x wider here.
//   shadow_kernel  one lane per (ray, bounce): the diffuse shadow ray to the microphone and the
//                  final Impulse (kernel.cpp:463-490).  nrays*nreflections independent any-hit
//                  queries: this is where the chip fills up.
//
// Every triangle test is the reference's Möller–Trumbore arithmetic (rvb_math.h); the BVH only
// prunes.  One wave per workgroup; the traversal stack lives in LDS ([entry][lane], no bank
// conflicts), nodes/triangles are read straight from L2 / Infinity Cache with 16-byte loads.
#include "kernels.h"
#include "rvb_math.h"

#define WAVE 64
#define NONE 0xFFFFFFFFu

namespace {

struct Hit { float t; uint32_t tri; uint32_t surface; };

__device__ __forceinline__ float clamp_inv(float d)
{
    float inv = 1.0f / d;                         // +-inf for d == 0
    return fminf(fmaxf(inv, -1e30f), 1e30f);      // keeps 0 * inf out of the slab test
}

// Closest hit (ANY = false): the brute-force winner of reference kernel.cpp:167-192.
// Any hit (ANY = true): is there a triangle with EPSILON < distance <= tmax — the negation of
// reference kernel.cpp:295 "(!inter.intersects) || inter.distance > mag".
template <bool ANY>
__device__ __forceinline__ bool traverse(const SceneDev & sc, const v3 o, const v3 d, const float tmax,
                                         uint32_t * __restrict__ stack /* this lane's column, stride WAVE */, Hit & hit)
{
    const float ix = clamp_inv(d.x), iy = clamp_inv(d.y), iz = clamp_inv(d.z);
    float best_t = ANY ? tmax : __builtin_inff();
    uint32_t best_i = NONE, best_s = 0;
    int sp = 0;
    uint32_t ref = 0;                             // root node
    for (;;) {
        while (!(ref & RVB_BVH_LEAF)) {
            const float4 * n = reinterpret_cast<const float4 *>(sc.nodes + ref);
            const float4 lox = n[0], loy = n[1], loz = n[2], hix = n[3], hiy = n[4], hiz = n[5];
            const uint4 ch = reinterpret_cast<const uint4 *>(n)[6];
            const float limit = best_t + (sc.cull_abs + sc.cull_rel * best_t);
            float key[4];
            uint32_t cref[4] = {ch.x, ch.y, ch.z, ch.w};
            const float lx[4] = {lox.x, lox.y, lox.z, lox.w}, ly[4] = {loy.x, loy.y, loy.z, loy.w}, lz[4] = {loz.x, loz.y, loz.z, loz.w};
            const float hx[4] = {hix.x, hix.y, hix.z, hix.w}, hy[4] = {hiy.x, hiy.y, hiy.z, hiy.w}, hz[4] = {hiz.x, hiz.y, hiz.z, hiz.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float tx0 = (lx[c] - o.x) * ix, tx1 = (hx[c] - o.x) * ix;
                float ty0 = (ly[c] - o.y) * iy, ty1 = (hy[c] - o.y) * iy;
                float tz0 = (lz[c] - o.z) * iz, tz1 = (hz[c] - o.z) * iz;
                float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
                float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
                bool ok = cref[c] != RVB_BVH_EMPTY && tn <= tf && tf >= -sc.cull_abs && tn <= limit;
                key[c] = ok ? tn : __builtin_inff();
                if (!ok) cref[c] = NONE;
            }
            // sort the four children by entry distance (5-comparator network)
#define RVB_CSWAP(a, b) { if (key[b] < key[a]) { float tk = key[a]; key[a] = key[b]; key[b] = tk; uint32_t tr = cref[a]; cref[a] = cref[b]; cref[b] = tr; } }
            RVB_CSWAP(0, 1) RVB_CSWAP(2, 3) RVB_CSWAP(0, 2) RVB_CSWAP(1, 3) RVB_CSWAP(1, 2)
#undef RVB_CSWAP
            // misses carry key = inf and sort last; push far-to-near, continue with the nearest
            if (cref[3] != NONE) { stack[sp * WAVE] = cref[3]; ++sp; }
            if (cref[2] != NONE) { stack[sp * WAVE] = cref[2]; ++sp; }
            if (cref[1] != NONE) { stack[sp * WAVE] = cref[1]; ++sp; }
            if (cref[0] != NONE) {
                ref = cref[0];
            } else if (sp > 0) {
                --sp;
                ref = stack[sp * WAVE];
            } else {
                ref = NONE;                       // has the LEAF bit: leaves the inner loop
            }
        }
        if (ref == NONE)
            break;
        const uint32_t first = ref & 0x0FFFFFFFu;
        const uint32_t count = ((ref >> 28) & 7u) + 1u;
        for (uint32_t j = 0; j < count; ++j) {
            const float4 * tp = reinterpret_cast<const float4 *>(sc.tris + first + j);
            const float4 a = tp[0], b = tp[1], c = tp[2];
            const v3 v0 = mk3(a.x, a.y, a.z), e0 = mk3(a.w, b.x, b.y), e1 = mk3(b.z, b.w, c.x);
            const float dist = mt_intersect(v0, e0, e1, o, d);
            const uint32_t idx = __float_as_uint(c.y);
            if (ANY) {
                if (dist > RVB_EPSILON && dist <= tmax)
                    return true;
            } else {
                // kernel.cpp:180-188 — smallest distance wins, equal distances go to the lower index
                if (dist > RVB_EPSILON && (best_i == NONE || dist < best_t || (dist == best_t && idx < best_i))) {
                    best_t = dist;
                    best_i = idx;
                    best_s = __float_as_uint(c.z);
                }
            }
        }
        if (sp > 0) {
            --sp;
            ref = stack[sp * WAVE];
        } else {
            break;
        }
    }
    if (ANY)
        return false;
    hit.t = best_t;
    hit.tri = best_i;
    hit.surface = best_s;
    return best_i != NONE;
}

// reference kernel.cpp:274-296 (point_intersection): is `point` visible from `begin`
__device__ __forceinline__ bool point_visible(const SceneDev & sc, v3 begin, v3 point, uint32_t * stack, float & mag)
{
    const v3 b2p = point - begin;
    mag = length3(b2p);
    const v3 dir = normalize3(b2p);
    Hit h;
    return !traverse<true>(sc, begin, dir, mag, stack, h);
}

__device__ __forceinline__ v3 ld3(const float * p) { return mk3(p[0], p[1], p[2]); }

// ------------------------------------------------------------------------------------------------
// Work record left by path_kernel in impulses[ray*nrefl + bounce] (64 B, four 16-byte stores):
//   [0..7]  newVol = -volume * specular                      (kernel.cpp:461)
//   [8..10] intersection, [11] DIFF = |dot(normal, dir)|      (kernel.cpp:459, :478)
//   [12]    newDist (cumulative path length)                  (kernel.cpp:460)
//   [13]    surface index, [14] triangle index, [15] 1 = record valid
// shadow_kernel turns it into the final Impulse in place.
__global__ __launch_bounds__(WAVE) void path_kernel(TraceArgs a)
{
    __shared__ uint32_t stack_lds[RVB_BVH_STACK * WAVE];
    const uint32_t lane = threadIdx.x;
    const uint64_t ray = (uint64_t) blockIdx.x * WAVE + lane;
    if (ray >= a.nrays)
        return;
    uint32_t * stack = stack_lds + lane;

    const float4 d4 = a.directions[ray];
    v3 o = ld3(a.source);
    v3 d = mk3(d4.x, d4.y, d4.z);
    float distance = 0.0f;
    float vol[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) vol[b] = 1.0f;

    float4 * out = reinterpret_cast<float4 *>(a.impulses + ray * a.nreflections);
    uint32_t index = 0;
    for (; index < a.nreflections; ++index) {
        Hit h;
        if (!traverse<false>(a.scene, o, d, 0.0f, stack, h))
            break;                                                   // kernel.cpp:372-375
        const float4 sh = reinterpret_cast<const float4 *>(a.scene.shade)[h.tri];
        const v3 normal = mk3(sh.x, sh.y, sh.z);
        const float4 * sp = reinterpret_cast<const float4 *>(a.scene.surfaces + h.surface);
        const float4 s0 = sp[0], s1 = sp[1];
        const v3 p = o + d * h.t;                                    // kernel.cpp:459
        const float new_dist = distance + h.t;                       // kernel.cpp:460
        vol[0] = -vol[0] * s0.x; vol[1] = -vol[1] * s0.y; vol[2] = -vol[2] * s0.z; vol[3] = -vol[3] * s0.w;
        vol[4] = -vol[4] * s1.x; vol[5] = -vol[5] * s1.y; vol[6] = -vol[6] * s1.z; vol[7] = -vol[7] * s1.w;
        const float diff = fabsf(dot3(normal, d));                   // kernel.cpp:478
        out[4 * index + 0] = make_float4(vol[0], vol[1], vol[2], vol[3]);
        out[4 * index + 1] = make_float4(vol[4], vol[5], vol[6], vol[7]);
        out[4 * index + 2] = make_float4(p.x, p.y, p.z, diff);
        out[4 * index + 3] = make_float4(new_dist, __uint_as_float(h.surface), __uint_as_float(h.tri), __uint_as_float(1u));
        if (index < RVB_NUM_IMAGE_SOURCE - 1)
            a.early[ray * (RVB_NUM_IMAGE_SOURCE - 1) + index] = h.tri;
        d = reflect3(normal, d);                                     // kernel.cpp:492-499
        o = p;
        distance = new_dist;
    }
    atomicAdd(a.executed, (unsigned long long) index);
}

// reference kernel.cpp:243-265 (add_image) for a known-valid slot
__device__ __forceinline__ void make_image(const TraceArgs & a, v3 mic, v3 mic_reflection, v3 source,
                                           const float volume[8], rvb_impulse & out)
{
    const v3 diff = source - mic_reflection;
    const float dist = length3(diff);
#pragma unroll
    for (int b = 0; b < 8; ++b)
        out.volume[b] = volume[b] * (air_attenuation(dist, a.air[b]) * 1.0f);
    const v3 pos = mic + diff;
    out.position[0] = pos.x; out.position[1] = pos.y; out.position[2] = pos.z; out.position[3] = 0.0f;
    out.time = seconds_per_meter() * dist;
    out.pad_[0] = out.pad_[1] = out.pad_[2] = 0.0f;
}

__device__ __forceinline__ TriVerts load_corners(const SceneDev & sc, uint32_t tri)
{
    const float4 * c = reinterpret_cast<const float4 *>(sc.corners + tri);
    const float4 a = c[0], b = c[1], e = c[2];
    TriVerts t;
    t.v0 = mk3(a.x, a.y, a.z);
    t.v1 = mk3(a.w, b.x, b.y);
    t.v2 = mk3(b.z, b.w, e.x);
    return t;
}

__global__ __launch_bounds__(WAVE) void image_kernel(TraceArgs a)
{
    __shared__ uint32_t stack_lds[RVB_BVH_STACK * WAVE];
    const uint32_t lane = threadIdx.x;
    uint32_t * stack = stack_lds + lane;
    const uint64_t g = (uint64_t) blockIdx.x * WAVE + lane;
    const v3 mic = ld3(a.mic), source = ld3(a.source);

    if (g == 0) {
        // slot 0, the direct path (kernel.cpp:335-357): identical for every ray, computed once
        rvb_impulse direct;
        for (int b = 0; b < 8; ++b) direct.volume[b] = 0.0f;
        for (int b = 0; b < 4; ++b) direct.position[b] = 0.0f;
        direct.time = 0.0f;
        direct.pad_[0] = direct.pad_[1] = direct.pad_[2] = 0.0f;
        float mag;
        if (point_visible(a.scene, source, mic, stack, mag)) {
            float one[8] = {1, 1, 1, 1, 1, 1, 1, 1};
            make_image(a, mic, mic, source, one, direct);
        }
        *a.direct = direct;
    }

    const uint32_t per_ray = RVB_NUM_IMAGE_SOURCE - 1;
    if (g >= a.nrays * per_ray)
        return;
    const uint64_t ray = g / per_ray;
    const uint32_t index = (uint32_t) (g % per_ray);
    if (index >= a.nreflections)
        return;
    const uint32_t * early = a.early + ray * per_ray;
    const uint32_t tri_here = early[index];
    if (tri_here == NONE)
        return;                                   // the ray escaped before this bounce

    // kernel.cpp:381-394: mirror the triangle chain and the microphone
    TriVerts prev[RVB_NUM_IMAGE_SOURCE - 1];
    v3 mic_reflection = mic;
    for (uint32_t k = 0; k <= index; ++k) {
        TriVerts current = load_corners(a.scene, early[k]);
        for (uint32_t j = 0; j < k; ++j)
            mirror_verts(current, prev[j]);
        prev[k] = current;
        mirror_point(mic_reflection, current);
    }

    // kernel.cpp:396-429
    const v3 dir = normalize3(mic_reflection - source);
    bool intersects = true;
    v3 prev_intersection = source;
    for (uint32_t k = 0; k != index + 1 && intersects; ++k) {
        const float to_intersection = mt_intersect_verts(prev[k], source, dir);
        if (to_intersection <= RVB_EPSILON) {
            intersects = false;
            break;
        }
        v3 ip = source + dir * to_intersection;
        for (int l = (int) k - 1; l != -1; --l)
            mirror_point(ip, prev[l]);

        const v3 idir = normalize3(ip - prev_intersection);
        Hit h;
        const bool found = traverse<false>(a.scene, prev_intersection, idir, 0.0f, stack, h);
        const float hd = found ? h.t : 0.0f;                          // Intersection {0, 0, false}
        const v3 nip = prev_intersection + idir * hd;
        const bool lo = (nip.x - RVB_EPSILON < ip.x) && (nip.y - RVB_EPSILON < ip.y) && (nip.z - RVB_EPSILON < ip.z);
        const bool hi = (ip.x < nip.x + RVB_EPSILON) && (ip.y < nip.y + RVB_EPSILON) && (ip.z < nip.z + RVB_EPSILON);
        intersects = found && lo && hi;
        prev_intersection = ip;
    }
    if (intersects) {
        float mag;
        intersects = point_visible(a.scene, prev_intersection, mic, stack, mag);   // kernel.cpp:431-440
    }
    if (!intersects)
        return;

    // kernel.cpp:442-456: the ray's volume BEFORE this bounce's surface is applied
    float volume[8];
    if (index == 0) {
        for (int b = 0; b < 8; ++b) volume[b] = 1.0f;
    } else {
        const float4 * rec = reinterpret_cast<const float4 *>(a.impulses + ray * a.nreflections + (index - 1));
        const float4 v0 = rec[0], v1 = rec[1];
        volume[0] = v0.x; volume[1] = v0.y; volume[2] = v0.z; volume[3] = v0.w;
        volume[4] = v1.x; volume[5] = v1.y; volume[6] = v1.z; volume[7] = v1.w;
    }
    rvb_image_candidate c;
    c.ray = a.ray_offset + ray;
    c.slot = index + 1;
    c.index = tri_here + 1;
    make_image(a, mic, mic_reflection, source, volume, c.impulse);
    const uint32_t at = atomicAdd(a.candidate_count, 1u);
    a.candidates[at] = c;
}

__global__ __launch_bounds__(WAVE) void shadow_kernel(TraceArgs a)
{
    __shared__ uint32_t stack_lds[RVB_BVH_STACK * WAVE];
    const uint32_t lane = threadIdx.x;
    uint32_t * stack = stack_lds + lane;
    const uint64_t total = a.nrays * (uint64_t) a.nreflections;
    const v3 mic = ld3(a.mic);
    for (uint64_t g = (uint64_t) blockIdx.x * WAVE + lane; g < total; g += (uint64_t) gridDim.x * WAVE) {
        float4 * rec = reinterpret_cast<float4 *>(a.impulses + g);
        const float4 r3 = rec[3];
        if (__float_as_uint(r3.w) != 1u)
            continue;                             // ray had already escaped: slot keeps its zero fill
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        const v3 p = mk3(r2.x, r2.y, r2.z);
        const float diff = r2.w;
        const float new_dist = r3.x;
        const uint32_t surface = __float_as_uint(r3.y);

        float mag;
        const bool visible = point_visible(a.scene, p, mic, stack, mag);        // kernel.cpp:463-469
        const float dist = visible ? new_dist + mag : 0.0f;                      // kernel.cpp:471
        float4 o0 = make_float4(0, 0, 0, 0), o1 = make_float4(0, 0, 0, 0);
        if (visible) {
            const float4 * sp = reinterpret_cast<const float4 *>(a.scene.surfaces + surface);
            const float4 d0 = sp[2], d1 = sp[3];                                  // diffuse coefficients
            // kernel.cpp:480-485: newVol * attenuation * diffuse * DIFF, left to right
            o0.x = ((r0.x * (air_attenuation(dist, a.air[0]) * 1.0f)) * d0.x) * diff;
            o0.y = ((r0.y * (air_attenuation(dist, a.air[1]) * 1.0f)) * d0.y) * diff;
            o0.z = ((r0.z * (air_attenuation(dist, a.air[2]) * 1.0f)) * d0.z) * diff;
            o0.w = ((r0.w * (air_attenuation(dist, a.air[3]) * 1.0f)) * d0.w) * diff;
            o1.x = ((r1.x * (air_attenuation(dist, a.air[4]) * 1.0f)) * d1.x) * diff;
            o1.y = ((r1.y * (air_attenuation(dist, a.air[5]) * 1.0f)) * d1.y) * diff;
            o1.z = ((r1.z * (air_attenuation(dist, a.air[6]) * 1.0f)) * d1.z) * diff;
            o1.w = ((r1.w * (air_attenuation(dist, a.air[7]) * 1.0f)) * d1.w) * diff;
        }
        rec[0] = o0;
        rec[1] = o1;
        rec[2] = make_float4(p.x, p.y, p.z, 0.0f);
        rec[3] = make_float4(seconds_per_meter() * dist, 0.0f, 0.0f, 0.0f);      // kernel.cpp:489
    }
}

}  // namespace

void rvb_launch_path(const TraceArgs & a, hipStream_t s)
{
    if (a.nrays == 0) return;
    const unsigned blocks = (unsigned) ((a.nrays + WAVE - 1) / WAVE);
    hipLaunchKernelGGL(path_kernel, dim3(blocks), dim3(WAVE), 0, s, a);
}

void rvb_launch_images(const TraceArgs & a, hipStream_t s)
{
    const uint64_t work = a.nrays * (RVB_NUM_IMAGE_SOURCE - 1);
    const unsigned blocks = (unsigned) ((work + WAVE - 1) / WAVE);
    hipLaunchKernelGGL(image_kernel, dim3(blocks ? blocks : 1), dim3(WAVE), 0, s, a);
}

void rvb_launch_shadow(const TraceArgs & a, hipStream_t s)
{
    const uint64_t total = a.nrays * (uint64_t) a.nreflections;
    if (total == 0) return;
    uint64_t blocks = (total + WAVE - 1) / WAVE;
    if (blocks > 256u * 64u) blocks = 256u * 64u;     // grid-stride beyond 64 waves per CU
    hipLaunchKernelGGL(shadow_kernel, dim3((unsigned) blocks), dim3(WAVE), 0, s, a);
}
