// trace_kernels.hip — the per-ray trace of reference rayverb/kernel.cpp:304-503 (kernel
// `raytrace`), re-organised for CDNA4 as three kernels over one 4-wide BVH:
//
//   path_kernel    the inherently sequential chain closest hit -> reflect (kernel.cpp:359-375,
//                  :459-461, :478, :492-501).  Latency-bound: FOUR LANES PER RAY, so that 100k rays
//                  are 6250 waves instead of 1563 and the chip has enough waves to hide the
//                  dependent node fetches.  Per bounce it leaves a 64-byte work record in the
//                  ray's Impulse slot.
//   image_kernel   one lane per (ray, bounce < 9): image-source validation (kernel.cpp:379-457).
//                  The reference does this inside the ray's loop; its inputs are only the
//                  triangles the ray hit so far, so it parallelises 9x wider here.
//   shadow_kernel  four lanes per (ray, bounce): the diffuse shadow ray to the microphone and the
//                  final Impulse (kernel.cpp:463-490).  nrays*nreflections independent any-hit
//                  queries: this is where the chip fills up.
//
// Quad-cooperative traversal: the four lanes of a quad own the four children of a node (one
// contiguous 128-byte line per visit, two 16-byte loads per lane) and the up-to-four triangles
// of a leaf; they combine results with DPP quad_perm moves, never through memory.  The per-ray
// stack lives in LDS, 4 bytes per entry.  Every triangle test is the reference's Möller–Trumbore
// arithmetic (rvb_math.h); the BVH only prunes, so a query returns the brute-force answer.
#include "kernels.h"
#include "rvb_math.h"

#define WAVE 64
#define QUADS_PER_BLOCK 16          // rays (or records) per 64-lane workgroup in the quad kernels
#define NONE 0xFFFFFFFFu

namespace {

struct Hit { float t; uint32_t tri; };

__device__ __forceinline__ float clamp_inv(float d)
{
    float inv = 1.0f / d;                         // +-inf for d == 0
    return fminf(fmaxf(inv, -1e30f), 1e30f);      // keeps 0 * inf out of the slab test
}

// ---- DPP helpers: data movement inside a quad (lanes 4k .. 4k+3) --------------------------------
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }
#define QP_SWAP1 0xB1     // quad_perm [1,0,3,2]
#define QP_SWAP2 0x4E     // quad_perm [2,3,0,1]
#define QP_BCAST(k) ((k) * 0x55)
template <int K> __device__ __forceinline__ float quad_bcast_f(float v) { return dpp_f<QP_BCAST(K)>(v); }
template <int K> __device__ __forceinline__ uint32_t quad_bcast_u(uint32_t v) { return dpp_u<QP_BCAST(K)>(v); }

// 4-bit mask of `pred` over this lane's quad
__device__ __forceinline__ uint32_t quad_ballot(bool pred)
{
    const unsigned long long m = __ballot(pred);
    return (uint32_t) (m >> (threadIdx.x & 60u)) & 0xFu;
}

// Slab test of one child box.  Boxes are padded by the builder (BuiltScene::pad), limit carries the
// cull slack, so the test is conservative with respect to the float triangle test.
__device__ __forceinline__ bool slab(const float4 a, const float4 b, const v3 o, const float ix, const float iy, const float iz,
                                     const float limit, const float cull_abs, float & tn)
{
    const float tx0 = (a.x - o.x) * ix, tx1 = (a.w - o.x) * ix;
    const float ty0 = (a.y - o.y) * iy, ty1 = (b.x - o.y) * iy;
    const float tz0 = (a.z - o.z) * iz, tz1 = (b.y - o.z) * iz;
    tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    return __float_as_uint(b.z) != RVB_BVH_EMPTY && tn <= tf && tf >= -cull_abs && tn <= limit;
}

// Closest hit (ANY = false): the brute-force winner of reference kernel.cpp:167-192.
// Any hit (ANY = true): is there a triangle with EPSILON < distance <= tmax — the negation of
// reference kernel.cpp:295 "(!inter.intersects) || inter.distance > mag".
// All four lanes of the quad call this with identical o, d, tmax and get identical results.
// stack: this ray's column of the LDS stack, entries QUADS_PER_BLOCK words apart.
template <bool ANY>
__device__ __forceinline__ bool traverse_quad(const SceneDev & sc, const v3 o, const v3 d, const float tmax,
                                              uint32_t * __restrict__ stack, Hit & hit)
{
    const uint32_t c = threadIdx.x & 3u;          // the child / leaf triangle this lane owns
    const float ix = clamp_inv(d.x), iy = clamp_inv(d.y), iz = clamp_inv(d.z);
    float best_t = ANY ? tmax : __builtin_inff();
    uint32_t best_i = NONE;
    uint32_t sp = 0;
    uint32_t ref = 0;                             // root node
    for (;;) {
        while (!(ref & RVB_BVH_LEAF)) {
            const float4 * n = reinterpret_cast<const float4 *>(sc.nodes + ref) + 2 * c;
            const float4 a = n[0], b = n[1];
            const float limit = best_t * (1.0f + sc.cull_rel) + sc.cull_abs;
            float tn;
            const bool ok = slab(a, b, o, ix, iy, iz, limit, sc.cull_abs, tn);
            const uint32_t cref = __float_as_uint(b.z);
            const uint32_t hits = quad_ballot(ok);
            if (hits == 0) {
                if (sp > 0) { --sp; ref = stack[sp * QUADS_PER_BLOCK]; } else ref = NONE;
                continue;
            }
            uint32_t winner;
            if (ANY) {
                winner = __ffs(hits) - 1;         // any-hit does not care about visiting order
            } else {
                const float key = ok ? tn : __builtin_inff();
                const float k1 = fminf(key, dpp_f<QP_SWAP1>(key));
                const float kmin = fminf(k1, dpp_f<QP_SWAP2>(k1));
                winner = __ffs(quad_ballot(ok && key == kmin)) - 1;   // nearest child first
            }
            const uint32_t rest = hits & ~(1u << winner);
            if (ok && c != winner)
                stack[(sp + __popc(rest & ((1u << c) - 1u))) * QUADS_PER_BLOCK] = cref;
            sp += __popc(rest);
            const uint32_t r0 = quad_bcast_u<0>(cref), r1 = quad_bcast_u<1>(cref), r2 = quad_bcast_u<2>(cref), r3 = quad_bcast_u<3>(cref);
            ref = winner == 0 ? r0 : winner == 1 ? r1 : winner == 2 ? r2 : r3;
        }
        if (ref == NONE)
            break;
        const uint32_t first = ref & 0x0FFFFFFFu;
        const uint32_t count = ((ref >> 28) & 7u) + 1u;
        float dist = 0.0f;
        uint32_t idx = NONE;
        if (c < count) {
            const float4 * tp = reinterpret_cast<const float4 *>(sc.tris + first + c);
            const float4 ta = tp[0], tb = tp[1], tc = tp[2];
            dist = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d);
            idx = __float_as_uint(tc.y);
        }
        if (ANY) {
            if (quad_ballot(c < count && dist > RVB_EPSILON && dist <= tmax) != 0)
                return true;
        } else {
            // kernel.cpp:180-188 — smallest distance wins, equal distances go to the lower index.
            // Lexicographic (distance, index) minimum over the quad's valid lanes.
            const bool valid = c < count && dist > RVB_EPSILON;
            float rd = valid ? dist : __builtin_inff();
            uint32_t ri = valid ? idx : NONE;
            {
                const float od = dpp_f<QP_SWAP1>(rd);
                const uint32_t oi = dpp_u<QP_SWAP1>(ri);
                if (od < rd || (od == rd && oi < ri)) { rd = od; ri = oi; }
            }
            {
                const float od = dpp_f<QP_SWAP2>(rd);
                const uint32_t oi = dpp_u<QP_SWAP2>(ri);
                if (od < rd || (od == rd && oi < ri)) { rd = od; ri = oi; }
            }
            if (ri != NONE && (best_i == NONE || rd < best_t || (rd == best_t && ri < best_i))) {
                best_t = rd;
                best_i = ri;
            }
        }
        if (sp > 0) { --sp; ref = stack[sp * QUADS_PER_BLOCK]; } else break;
    }
    if (ANY)
        return false;
    hit.t = best_t;
    hit.tri = best_i;
    return best_i != NONE;
}

// One-lane-per-query traversal (image_kernel): same tests, same rule, the lane walks all four
// children itself.  stack: this lane's column, entries WAVE words apart.
template <bool ANY>
__device__ __forceinline__ bool traverse_lane(const SceneDev & sc, const v3 o, const v3 d, const float tmax,
                                              uint32_t * __restrict__ stack, Hit & hit)
{
    const float ix = clamp_inv(d.x), iy = clamp_inv(d.y), iz = clamp_inv(d.z);
    float best_t = ANY ? tmax : __builtin_inff();
    uint32_t best_i = NONE;
    int sp = 0;
    uint32_t ref = 0;
    for (;;) {
        while (!(ref & RVB_BVH_LEAF)) {
            const float4 * n = reinterpret_cast<const float4 *>(sc.nodes + ref);
            const float limit = best_t * (1.0f + sc.cull_rel) + sc.cull_abs;
            float key[4];
            uint32_t cref[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 a = n[2 * c], b = n[2 * c + 1];
                float tn;
                const bool ok = slab(a, b, o, ix, iy, iz, limit, sc.cull_abs, tn);
                key[c] = ok ? tn : __builtin_inff();
                cref[c] = ok ? __float_as_uint(b.z) : NONE;
            }
#define RVB_CSWAP(a, b) { if (key[b] < key[a]) { float tk = key[a]; key[a] = key[b]; key[b] = tk; uint32_t tr = cref[a]; cref[a] = cref[b]; cref[b] = tr; } }
            if (!ANY) {
                RVB_CSWAP(0, 1) RVB_CSWAP(2, 3) RVB_CSWAP(0, 2) RVB_CSWAP(1, 3) RVB_CSWAP(1, 2)
            }
#undef RVB_CSWAP
            if (cref[3] != NONE) { stack[sp * WAVE] = cref[3]; ++sp; }
            if (cref[2] != NONE) { stack[sp * WAVE] = cref[2]; ++sp; }
            if (cref[1] != NONE) { stack[sp * WAVE] = cref[1]; ++sp; }
            if (cref[0] != NONE) {
                ref = cref[0];
            } else if (sp > 0) {
                --sp;
                ref = stack[sp * WAVE];
            } else {
                ref = NONE;
            }
        }
        if (ref == NONE)
            break;
        const uint32_t first = ref & 0x0FFFFFFFu;
        const uint32_t count = ((ref >> 28) & 7u) + 1u;
        for (uint32_t j = 0; j < count; ++j) {
            const float4 * tp = reinterpret_cast<const float4 *>(sc.tris + first + j);
            const float4 ta = tp[0], tb = tp[1], tc = tp[2];
            const float dist = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d);
            const uint32_t idx = __float_as_uint(tc.y);
            if (ANY) {
                if (dist > RVB_EPSILON && dist <= tmax)
                    return true;
            } else if (dist > RVB_EPSILON && (best_i == NONE || dist < best_t || (dist == best_t && idx < best_i))) {
                best_t = dist;
                best_i = idx;
            }
        }
        if (sp > 0) { --sp; ref = stack[sp * WAVE]; } else break;
    }
    if (ANY)
        return false;
    hit.t = best_t;
    hit.tri = best_i;
    return best_i != NONE;
}

__device__ __forceinline__ v3 ld3(const float * p) { return mk3(p[0], p[1], p[2]); }

// ------------------------------------------------------------------------------------------------
// Work record left by path_kernel in impulses[ray*nrefl + bounce] (64 B; quad lane c stores chunk c):
//   chunk 0,1  newVol = -volume * specular                        (kernel.cpp:461)
//   chunk 2    intersection.xyz, DIFF = |dot(normal, dir)|         (kernel.cpp:459, :478)
//   chunk 3    newDist, surface index, triangle index, 1 = valid   (kernel.cpp:460)
// shadow_kernel turns it into the final Impulse in place.
__global__ __launch_bounds__(WAVE) void path_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][QUADS_PER_BLOCK]
    const uint32_t c = threadIdx.x & 3u;
    const uint32_t q = threadIdx.x >> 2;
    const uint64_t ray = (uint64_t) blockIdx.x * QUADS_PER_BLOCK + q;
    if (ray >= a.nrays)
        return;                                   // whole quads leave together
    uint32_t * stack = stack_lds + q;

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
        if (!traverse_quad<false>(a.scene, o, d, 0.0f, stack, h))
            break;                                                   // kernel.cpp:372-375
        const float4 sh = reinterpret_cast<const float4 *>(a.scene.shade)[h.tri];
        const v3 normal = mk3(sh.x, sh.y, sh.z);
        const uint32_t surface = __float_as_uint(sh.w);
        const float4 * sp = reinterpret_cast<const float4 *>(a.scene.surfaces + surface);
        const float4 s0 = sp[0], s1 = sp[1];
        const v3 p = o + d * h.t;                                    // kernel.cpp:459
        const float new_dist = distance + h.t;                       // kernel.cpp:460
        vol[0] = -vol[0] * s0.x; vol[1] = -vol[1] * s0.y; vol[2] = -vol[2] * s0.z; vol[3] = -vol[3] * s0.w;
        vol[4] = -vol[4] * s1.x; vol[5] = -vol[5] * s1.y; vol[6] = -vol[6] * s1.z; vol[7] = -vol[7] * s1.w;
        const float diff = fabsf(dot3(normal, d));                   // kernel.cpp:478
        float4 chunk;
        if (c == 0) chunk = make_float4(vol[0], vol[1], vol[2], vol[3]);
        else if (c == 1) chunk = make_float4(vol[4], vol[5], vol[6], vol[7]);
        else if (c == 2) chunk = make_float4(p.x, p.y, p.z, diff);
        else chunk = make_float4(new_dist, __uint_as_float(surface), __uint_as_float(h.tri), __uint_as_float(1u));
        out[4 * index + c] = chunk;
        if (c == 0 && index < RVB_NUM_IMAGE_SOURCE - 1)
            a.early[ray * (RVB_NUM_IMAGE_SOURCE - 1) + index] = h.tri;
        d = reflect3(normal, d);                                     // kernel.cpp:492-499
        o = p;
        distance = new_dist;
    }
    if (c == 0)
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

// reference kernel.cpp:274-296 (point_intersection), one lane
__device__ __forceinline__ bool point_visible_lane(const SceneDev & sc, v3 begin, v3 point, uint32_t * stack)
{
    const v3 b2p = point - begin;
    const float mag = length3(b2p);
    Hit h;
    return !traverse_lane<true>(sc, begin, normalize3(b2p), mag, stack, h);
}

__global__ __launch_bounds__(WAVE) void image_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][WAVE]
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
        if (point_visible_lane(a.scene, source, mic, stack)) {
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
        const bool found = traverse_lane<false>(a.scene, prev_intersection, idir, 0.0f, stack, h);
        const float hd = found ? h.t : 0.0f;                          // Intersection {0, 0, false}
        const v3 nip = prev_intersection + idir * hd;
        const bool lo = (nip.x - RVB_EPSILON < ip.x) && (nip.y - RVB_EPSILON < ip.y) && (nip.z - RVB_EPSILON < ip.z);
        const bool hi = (ip.x < nip.x + RVB_EPSILON) && (ip.y < nip.y + RVB_EPSILON) && (ip.z < nip.z + RVB_EPSILON);
        intersects = found && lo && hi;
        prev_intersection = ip;
    }
    if (intersects)
        intersects = point_visible_lane(a.scene, prev_intersection, mic, stack);   // kernel.cpp:431-440
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
    rvb_image_candidate cand;
    cand.ray = a.ray_offset + ray;
    cand.slot = index + 1;
    cand.index = tri_here + 1;
    make_image(a, mic, mic_reflection, source, volume, cand.impulse);
    const uint32_t at = atomicAdd(a.candidate_count, 1u);
    a.candidates[at] = cand;
}

__global__ __launch_bounds__(WAVE) void shadow_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][QUADS_PER_BLOCK]
    const uint32_t c = threadIdx.x & 3u;
    const uint32_t q = threadIdx.x >> 2;
    uint32_t * stack = stack_lds + q;
    const uint64_t total = a.nrays * (uint64_t) a.nreflections;
    const v3 mic = ld3(a.mic);
    // a.air[] for this lane's four bands (lanes 0/1 of the quad finish bands 0-3 / 4-7)
    const float air0 = a.air[(c & 1u) * 4 + 0], air1 = a.air[(c & 1u) * 4 + 1];
    const float air2 = a.air[(c & 1u) * 4 + 2], air3 = a.air[(c & 1u) * 4 + 3];
    for (uint64_t g = (uint64_t) blockIdx.x * QUADS_PER_BLOCK + q; g < total; g += (uint64_t) gridDim.x * QUADS_PER_BLOCK) {
        float4 * rec = reinterpret_cast<float4 *>(a.impulses + g);
        const float4 mine = rec[c];               // the quad reads the 64-byte record as one line
        // chunk 3 = (newDist, surface, triangle, valid); chunk 2 = (intersection, DIFF)
        const uint32_t valid = quad_bcast_u<3>(__float_as_uint(mine.w));
        if (valid != 1u)
            continue;                             // ray had already escaped: slot keeps its zero fill
        const float new_dist = quad_bcast_f<3>(mine.x);
        const uint32_t surface = quad_bcast_u<3>(__float_as_uint(mine.y));
        const v3 p = mk3(quad_bcast_f<2>(mine.x), quad_bcast_f<2>(mine.y), quad_bcast_f<2>(mine.z));
        const float diff = quad_bcast_f<2>(mine.w);

        // kernel.cpp:463-469 point_intersection(intersection, mic)
        const v3 b2p = mic - p;
        const float mag = length3(b2p);
        Hit h;
        const bool visible = !traverse_quad<true>(a.scene, p, normalize3(b2p), mag, stack, h);
        const float dist = visible ? new_dist + mag : 0.0f;          // kernel.cpp:471
        float4 o = make_float4(0, 0, 0, 0);
        if (c < 2) {
            if (visible) {
                const float4 dc = reinterpret_cast<const float4 *>(a.scene.surfaces + surface)[2 + c];   // diffuse
                // kernel.cpp:480-485: newVol * attenuation * diffuse * DIFF, left to right
                o.x = ((mine.x * (air_attenuation(dist, air0) * 1.0f)) * dc.x) * diff;
                o.y = ((mine.y * (air_attenuation(dist, air1) * 1.0f)) * dc.y) * diff;
                o.z = ((mine.z * (air_attenuation(dist, air2) * 1.0f)) * dc.z) * diff;
                o.w = ((mine.w * (air_attenuation(dist, air3) * 1.0f)) * dc.w) * diff;
            }
        } else if (c == 2) {
            o = make_float4(p.x, p.y, p.z, 0.0f);
        } else {
            o.x = seconds_per_meter() * dist;                        // kernel.cpp:489
        }
        rec[c] = o;
    }
}

}  // namespace

void rvb_launch_path(const TraceArgs & a, hipStream_t s)
{
    if (a.nrays == 0) return;
    const unsigned blocks = (unsigned) ((a.nrays + QUADS_PER_BLOCK - 1) / QUADS_PER_BLOCK);
    hipLaunchKernelGGL(path_kernel, dim3(blocks), dim3(WAVE), a.stack_entries * QUADS_PER_BLOCK * sizeof(uint32_t), s, a);
}

void rvb_launch_images(const TraceArgs & a, hipStream_t s)
{
    const uint64_t work = a.nrays * (RVB_NUM_IMAGE_SOURCE - 1);
    const unsigned blocks = (unsigned) ((work + WAVE - 1) / WAVE);
    hipLaunchKernelGGL(image_kernel, dim3(blocks ? blocks : 1), dim3(WAVE), a.stack_entries * WAVE * sizeof(uint32_t), s, a);
}

void rvb_launch_shadow(const TraceArgs & a, hipStream_t s)
{
    const uint64_t total = a.nrays * (uint64_t) a.nreflections;
    if (total == 0) return;
    uint64_t blocks = (total + QUADS_PER_BLOCK - 1) / QUADS_PER_BLOCK;
    if (blocks > 256u * 256u) blocks = 256u * 256u;   // grid-stride beyond 256 single-wave workgroups per CU
    hipLaunchKernelGGL(shadow_kernel, dim3((unsigned) blocks), dim3(WAVE), a.stack_entries * QUADS_PER_BLOCK * sizeof(uint32_t), s, a);
}
