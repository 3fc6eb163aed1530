// trace_kernels.hip — the per-ray trace of reference rayverb/kernel.cpp:304-503 (kernel
// `raytrace`), re-organised for CDNA4 as three kernels over one 4-wide BVH:
//
//   path_kernel / path_pair_kernel
//                  the inherently sequential chain closest hit -> reflect (kernel.cpp:359-375,
//                  :459-461, :478, :492-501).  Latency-bound: SEVERAL LANES PER RAY — four (100k rays are
//                  6250 waves instead of 1563: enough waves to hide the dependent node fetches) or two
//                  (a quarter fewer instructions per bounce; chosen when the rays in flight fill the chip
//                  anyway, rvb_path_lanes_for).  Per bounce it leaves a 64-byte work record in the ray's
//                  Impulse slot.
//   image_plan_kernel / image_check_kernel
//                  image-source validation of a ray's first nine bounces (kernel.cpp:379-457): one lane per ray lists the
//                  (ray, bounce) pairs whose image ray crosses every mirrored triangle, four lanes per listed pair and query
//                  run the closest-hit / any-hit checks.  The inputs are only the triangles the ray hit, so this runs
//                  beside the record grouping instead of inside the ray's loop.
//   shadow_pair_kernel (shadow_kernel: the four-lane form, kept for measurements)
//                  two lanes per (ray, bounce): the diffuse shadow ray to the microphone and the
//                  final Impulse (kernel.cpp:463-490).  nrays*nreflections independent any-hit
//                  queries: this is where the chip fills up.
//
// Lane-cooperative traversal: the lanes of a ray own the four children of a node (one contiguous 64-byte
// half line per visit, 16-byte loads) and the up-to-four triangles of a leaf; they combine results with
// DPP quad_perm moves, never through memory.  (Nodes are 64 bytes: binary16 boxes rounded outward.)  The
// per-ray stack lives in LDS, 4 bytes per entry.  Every triangle test is the reference's Möller–Trumbore
// arithmetic (rvb_math.h); the BVH only prunes, so a query returns the brute-force answer.
#include "kernels.h"
#include "rvb_math.h"

#include <algorithm>
#include <cstdlib>

#ifndef RVB_PATH_JOBS
#define RVB_PATH_JOBS 2     // 2: majority-vote job loop (traverse_jobs_vote), 1: while-while job loop, 0: one query at a time
#endif
#ifndef RVB_PROBE_NO_STORES
#define RVB_PROBE_NO_STORES 0
#endif
#ifndef RVB_QUAD_SELECT
#define RVB_QUAD_SELECT 1      // slab_select (near / far plane by the direction's sign) in the four-lane path kernel as well
#endif
#ifndef RVB_SHADOW_JOBS
#define RVB_SHADOW_JOBS 0
#endif
#ifndef RVB_CYCLE_LEAF_NUM
#define RVB_CYCLE_LEAF_NUM 3       // a leaf step when NUM x (lanes at a leaf) >= DEN x (live lanes)
#define RVB_CYCLE_LEAF_DEN 1
#endif
#ifndef RVB_CYCLE_DONE_NUM
#define RVB_CYCLE_DONE_NUM 4       // a shading step when NUM x (lanes with a finished query) >= DEN x (live lanes)
#define RVB_CYCLE_DONE_DEN 1
#endif
#ifndef RVB_LDS_NODES
#define RVB_LDS_NODES 0        // experiment: top nodes of the BVH staged in LDS per workgroup (path_kernel); 21 = levels 0-2
#endif

// tools/isa_mix.py: -DRVB_ISA_MARKS=1 leaves comment lines in the ISA at the borders of the step kinds of the vote loops (never in the shipped build)
#ifndef RVB_ISA_MARKS
#define RVB_ISA_MARKS 0
#endif
#if RVB_ISA_MARKS
#define RVB_MARK(name) asm volatile("; RVB_MARK " name)
#else
#define RVB_MARK(name)
#endif

#define WAVE 64
#define QUADS_PER_BLOCK 16          // rays (or records) per 64-lane workgroup in the quad kernels
#define NONE 0xFFFFFFFFu

namespace {

// Diagnostic build only (-DRVB_STAMPS=1, never shipped): per-wave s_memtime shares of the traversal
// loop, written to a side buffer that no other code reads (cdna_hip_programming.md §7 "In-kernel stamps").
#ifndef RVB_STAMPS
#define RVB_STAMPS 0
#endif
#if RVB_STAMPS
#define STAMP(var) { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
struct Stamps {
    unsigned long long node_steps = 0, node_cycles = 0, leaf_steps = 0, leaf_cycles = 0, done_calls = 0, done_cycles = 0;
    unsigned long long quad_node_steps = 0, quad_leaf_steps = 0, t0 = 0;
};
#else
#define STAMP(var)
#endif

struct Hit { float t; uint32_t tri; };

// Streaming accesses to the 64-byte work records / Impulses (written once, read once by a later kernel).
// RVB_STREAM_STORE picks the cache policy of the stores: 0 = nt (stays in the XCD's L2 until evicted),
// 1 = sc1, 2 = sc0 sc1 (write-through, the line is DROPPED from L2 — MI355X_MICROARCH.md "stores of each
// flavour"), so that the 819 MB record stream does not push the ~6 MB scene out of the 4 MiB per-XCD L2s.
#ifndef RVB_STREAM_STORE
#define RVB_STREAM_STORE 0
#endif
typedef float nt_float4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t * lds_u32_ptr;
typedef __attribute__((address_space(3))) const nt_float4 * lds_float4_ptr;  // keeps ds_read: a generic pointer would load flat
__device__ __forceinline__ void store_stream(float4 * p, const float4 v)
{
    nt_float4 t = {v.x, v.y, v.z, v.w};
#if RVB_STREAM_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(t) : "memory");
#elif RVB_STREAM_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(t) : "memory");
#else
    __builtin_nontemporal_store(t, reinterpret_cast<nt_float4 *>(p));
#endif
}
__device__ __forceinline__ float4 load_stream(const float4 * p)
{
    const nt_float4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_float4 *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

// Inverse direction for the (conservative, padded) slab test only — never used by a triangle test,
// so the 1-ulp hardware reciprocal is enough.
__device__ __forceinline__ float clamp_inv(float d)
{
    float inv = __builtin_amdgcn_rcpf(d);         // +-inf for d == 0
    return fminf(fmaxf(inv, -1e30f), 1e30f);      // keeps 0 * inf out of the slab test
}

// ---- DPP helpers: data movement inside a quad (lanes 4k .. 4k+3) --------------------------------
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }
template <int CTRL> __device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v)
{
    return ((unsigned long long) dpp_u<CTRL>((uint32_t) (v >> 32)) << 32) | dpp_u<CTRL>((uint32_t) v);
}
#define QP_SWAP1 0xB1     // quad_perm [1,0,3,2]
#define QP_SWAP2 0x4E     // quad_perm [2,3,0,1]
#define QP_BCAST(k) ((k) * 0x55)
template <int K> __device__ __forceinline__ float quad_bcast_f(float v) { return dpp_f<QP_BCAST(K)>(v); }
template <int K> __device__ __forceinline__ uint32_t quad_bcast_u(uint32_t v) { return dpp_u<QP_BCAST(K)>(v); }

// byte offset of leaf-order triangle i < 2^24 (rvb_build_scene's limit): one full-rate 24-bit multiply
// (the 32-bit v_mul_lo_u32 the compiler picks for i * 48 is a quarter-rate instruction)
__device__ __forceinline__ uint32_t tri_byte_offset(uint32_t i) { return __umul24(i, (uint32_t) sizeof(BvhTri)); }

// does `pred` hold in any lane of this lane's quad?  Two DPP ORs (a 64-bit ballot masked per quad costs 64-bit VALU compares)
__device__ __forceinline__ bool quad_any(bool pred)
{
    uint32_t p = pred ? 1u : 0u;
    p |= (uint32_t) __builtin_amdgcn_mov_dpp((int) p, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    p |= (uint32_t) __builtin_amdgcn_mov_dpp((int) p, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    return p != 0;
}

// 4-bit mask of `pred` over this lane's quad
__device__ __forceinline__ uint32_t quad_ballot(bool pred)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);   // the condition mask itself, no 0/1 round trip through a VGPR
    return (uint32_t) (m >> (threadIdx.x & 60u)) & 0xFu;
}

// Slab test of one child box, t = lo*inv - o*inv as one FMA per plane.  A child record is 16 bytes:
// six binary16 planes rounded outward by the builder + the child reference.  Boxes are padded
// (BuiltScene::pad) and `limit` carries the cull slack, so the test is conservative with respect to
// the float triangle test (the FMA form moves a plane by <2e-3 of the padding).
// Folded: tn = max(entry, -cull_abs), tf = min(exit, limit); hit iff tn <= tf.  Empty child slots
// are rejected by their ref (minNum/maxNum would swallow a NaN box: max(NaN, -cull) = -cull).
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2_t as_half2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }

// `skip`: a child reference the query must not enter (the own-plane subtree of the triangle the ray starts on, TriShade in bvh.h;
// RVB_BVH_EMPTY = none, which doubles as the test for an empty slot).
__device__ __forceinline__ bool slab(const uint4 n, const float ix, const float iy, const float iz,
                                     const float oix, const float oiy, const float oiz,
                                     const float limit, const float neg_cull, const uint32_t skip, float & tn)
{
    const half2_t h0 = as_half2(n.x), h1 = as_half2(n.y), h2 = as_half2(n.z);   // (lo.x, hi.x) (lo.y, hi.y) (lo.z, hi.z)
    const float tx0 = fmaf((float) h0.x, ix, -oix), tx1 = fmaf((float) h0.y, ix, -oix);
    const float ty0 = fmaf((float) h1.x, iy, -oiy), ty1 = fmaf((float) h1.y, iy, -oiy);
    const float tz0 = fmaf((float) h2.x, iz, -oiz), tz1 = fmaf((float) h2.y, iz, -oiz);
    // The two folds with loop-invariant operands are written as instructions: fmaxf / fminf would first canonicalise
    // `neg_cull` and `limit` (values from another basic block are not known to be quiet) — two more VALU operations per
    // node step.  Neither is ever NaN; v_max / v_min return the other operand for a NaN box plane like fmaxf / fminf.
    float zn = fminf(tz0, tz1), zf = fmaxf(tz0, tz1);
    asm("v_max_f32 %0, %1, %2" : "=v"(zn) : "s"(neg_cull), "v"(zn));      // wave-uniform: stays in an SGPR
    asm("v_min_f32 %0, %1, %2" : "=v"(zf) : "v"(zf), "v"(limit));
    tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), zn);
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), zf);
    return tn <= tf && n.w != RVB_BVH_EMPTY && n.w != skip;
}

// The same test with the near / far plane of each axis SELECTED by the sign of the direction instead of computed as min / max of
// both products: one v_perm_b32 per axis swaps the halves of the (lo, hi) word when the ray runs towards -axis, after which the low
// half is the plane the ray meets first.  3 selects + 2 three-operand min / max replace 6 two-operand min / max and the EMPTY compare
// (an empty slot is an inverted infinite box: entry +inf, exit -inf).  The products are monotonic in the plane, so entry and exit
// are bit-identical to slab()'s.  sel*: slab_selector(inverse direction), 3 more registers per query — used by the two-lane
// kernels, whose register budget is not the 64 of the quad kernels.
__device__ __forceinline__ uint32_t slab_selector(float inv) { return inv < 0.0f ? 0x01000302u : 0x03020100u; }
__device__ __forceinline__ bool slab_select(const uint4 n, const float ix, const float iy, const float iz,
                                            const float oix, const float oiy, const float oiz,
                                            const uint32_t selx, const uint32_t sely, const uint32_t selz,
                                            const float limit, const float neg_cull, const uint32_t skip, float & tn)
{
    const half2_t hx = as_half2(__builtin_amdgcn_perm(n.x, n.x, selx)), hy = as_half2(__builtin_amdgcn_perm(n.y, n.y, sely)),
                  hz = as_half2(__builtin_amdgcn_perm(n.z, n.z, selz));      // (near, far) per axis
    const float nx = fmaf((float) hx.x, ix, -oix), fx = fmaf((float) hx.y, ix, -oix);
    const float ny = fmaf((float) hy.x, iy, -oiy), fy = fmaf((float) hy.y, iy, -oiy);
    const float nz = fmaf((float) hz.x, iz, -oiz), fz = fmaf((float) hz.y, iz, -oiz);
    float zn = nz, zf = fz;
    asm("v_max_f32 %0, %1, %2" : "=v"(zn) : "s"(neg_cull), "v"(zn));      // (written as instructions: see slab)
    asm("v_min_f32 %0, %1, %2" : "=v"(zf) : "v"(zf), "v"(limit));
    tn = fmaxf(fmaxf(nx, ny), zn);
    const float tf = fminf(fminf(fx, fy), zf);
    return tn <= tf && n.w != skip;
}

// Closest hit (ANY = false): the brute-force winner of reference kernel.cpp:167-192.
// Any hit (ANY = true): is there a triangle with EPSILON < distance <= tmax — the negation of
// reference kernel.cpp:295 "(!inter.intersects) || inter.distance > mag".
//
// Persistent job loop: a quad asks its Job for a query (job.next), traverses, hands the result
// back (job.done) and immediately asks for the next one, while the other quads of the wave keep
// traversing their own queries.  No quad ever waits for the slowest ray of its wave at a bounce /
// record boundary; the wave ends when every quad has run out of jobs.
//   bool Job::next(v3 & o, v3 & d, float & tmax)   set up the quad's next query, false = none left
//   void Job::done(bool hit, const Hit & h)          consume the result (quad-uniform control flow)
// stack: this quad's column of the LDS stack, entries QUADS_PER_BLOCK words apart.
template <bool ANY, class Job>
__device__ __forceinline__ void traverse_jobs(const SceneDev & sc, uint32_t * __restrict__ stack, Job & job)
{
    const uint32_t c = threadIdx.x & 3u;          // the child / leaf triangle this lane owns
    const uint32_t lane_base4 = (threadIdx.x & 60u) << 2;         // ds_bpermute address of the quad's lane 0
    const uint32_t lane_bit = 1u << c, lt_mask = lane_bit - 1u;
    const char * node_base = reinterpret_cast<const char *>(sc.nodes);   // wave-uniform: the load is base (SGPRs) + 32-bit lane offset
    const uint32_t child_off = 16u * c;
    const float neg_cull = -sc.cull_abs, cull_scale = 1.0f + sc.cull_rel;
    v3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
    float tmax = 0.0f;
    float ix = 0.0f, iy = 0.0f, iz = 0.0f, oix = 0.0f, oiy = 0.0f, oiz = 0.0f, best_t = 0.0f;
    uint32_t best_i = NONE, sp = 0, ref = 0;
#if RVB_STAMPS
    Stamps st;
    unsigned long long ta = 0, tb = 0;
    STAMP(st.t0)
#endif
    bool active = job.next(o, d, tmax);
#define RVB_RESET_QUERY()                                                         \
    {                                                                             \
        ix = clamp_inv(d.x); iy = clamp_inv(d.y); iz = clamp_inv(d.z);            \
        oix = o.x * ix; oiy = o.y * iy; oiz = o.z * iz;                           \
        best_t = ANY ? tmax : __builtin_inff();                                   \
        best_i = NONE; sp = 0; ref = 0;                                           \
    }
    if (active) RVB_RESET_QUERY()
    while (active) {
        while (!(ref & RVB_BVH_LEAF)) {
            STAMP(ta)
#if RVB_STAMPS
            st.quad_node_steps += (threadIdx.x & 3u) == 0 ? 1 : 0;
#endif
            const uint4 n = *reinterpret_cast<const uint4 *>(node_base + (ref | child_off));
            const float limit = fmaf(best_t, cull_scale, sc.cull_abs);
            float tn;
            const bool ok = slab(n, ix, iy, iz, oix, oiy, oiz, limit, neg_cull, job.skip_ref(), tn);
            const uint32_t cref = n.w;
            // key = entry distance (two mantissa bits traded for the lane id): the quad minimum names
            // the nearest hit child and the lane that owns it in two DPP steps
            uint32_t key = ok ? ((__float_as_uint(fmaxf(tn, 0.0f)) & ~3u) | c) : NONE;
            if (ANY) key = ok ? c : NONE;         // any-hit does not care about visiting order
            uint32_t kmin = min(key, dpp_u<QP_SWAP1>(key));
            kmin = min(kmin, dpp_u<QP_SWAP2>(kmin));
            if (kmin == NONE) {
                if (sp > 0) { --sp; ref = stack[sp * QUADS_PER_BLOCK]; } else ref = NONE;
#if RVB_STAMPS
                STAMP(tb)
                st.node_steps += 1; st.node_cycles += tb - ta;
#endif
                continue;
            }
            const uint32_t winner = kmin & 3u;
            // the quad's hit mask by two DPP ORs (a 64-bit ballot shifted down per quad costs a 64-bit VALU shift)
            uint32_t okmask = ok ? lane_bit : 0u;
            okmask |= dpp_u<QP_SWAP1>(okmask);
            okmask |= dpp_u<QP_SWAP2>(okmask);
            const uint32_t rest = okmask & ~(1u << winner);
            if (ok && c != winner)
                stack[(sp + __popc(rest & lt_mask)) * QUADS_PER_BLOCK] = cref;
            sp += __popc(rest);
            ref = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (lane_base4 + (winner << 2)), (int) cref);
#if RVB_STAMPS
            STAMP(tb)
            st.node_steps += 1; st.node_cycles += tb - ta;
#endif
        }
        STAMP(ta)
        bool finished = true, found = false;
        if (ref != NONE) {
            const uint32_t first = ref & 0x0FFFFFFFu;
            const uint32_t count = ((ref >> 28) & 7u) + 1u;
            float dist = 0.0f;
            uint32_t idx = NONE;
            if (c < count) {
                const float4 * tp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.tris) + tri_byte_offset(first + c));
                float4 ta = tp[0], tb = tp[1], tc = tp[2];
                // all three loads leave before the first use: without this the compiler sinks the v0 load below the
                // |det| test of mt_intersect and a leaf step pays two dependent round trips instead of one
                asm volatile("" : "+v"(ta.x), "+v"(tb.x), "+v"(tc.x));
                dist = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d);
                idx = __float_as_uint(tc.y);
            }
            if (ANY) {
                found = quad_any(c < count && dist > RVB_EPSILON && dist <= tmax);
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
            if (!found && sp > 0) {
                --sp;
                ref = stack[sp * QUADS_PER_BLOCK];
                finished = false;
            }
        }
#if RVB_STAMPS
        STAMP(tb)
        st.leaf_steps += 1; st.leaf_cycles += tb - ta;
        st.quad_leaf_steps += ((threadIdx.x & 3u) == 0 && ref != NONE) ? 1 : 0;
#endif
        if (finished) {
            STAMP(ta)
            Hit h;
            h.t = best_t;
            h.tri = best_i;
            job.done(ANY ? found : best_i != NONE, h);
            active = job.next(o, d, tmax);
            if (active) RVB_RESET_QUERY()
#if RVB_STAMPS
            STAMP(tb)
            st.done_calls += 1; st.done_cycles += tb - ta;
#endif
        }
    }
#if RVB_STAMPS
    if (sc.stamps) {
        STAMP(tb)
        // wave-level values are the maximum over lanes (a lane counts the wave steps it took part in)
        unsigned long long v[9] = {st.node_steps, st.node_cycles, st.leaf_steps, st.leaf_cycles, st.done_calls, st.done_cycles,
                                   tb - st.t0, st.quad_node_steps, st.quad_leaf_steps};
        for (int i = 0; i < 7; ++i) {
            unsigned long long m = v[i];
            for (int off = 32; off > 0; off >>= 1) { unsigned long long o2 = __shfl_xor(m, off); m = o2 > m ? o2 : m; }
            if ((threadIdx.x & 63u) == 0) atomicAdd(sc.stamps + i, m);
        }
        atomicAdd(sc.stamps + 7, v[7]);
        atomicAdd(sc.stamps + 8, v[8]);
        if ((threadIdx.x & 63u) == 0) atomicAdd(sc.stamps + 9, 1ull);
    }
#endif
#undef RVB_RESET_QUERY
}

// min of two unsigned 64-bit keys.  The compiler's form is v_cmp_lt_u64 -> VCC and two v_cndmask_b32 that read VCC; the SECOND
// select on one VCC value issues far slower than the first (tools/inst_probe.hip "cmpsel2_vcc": 3.0 ns against 0.9 ns for the
// same select on an SGPR-pair mask at 8 waves per SIMD, and 5-10x that at low occupancy).  Here the mask lives in an SGPR pair.
#ifndef RVB_MIN64_SGPR
#define RVB_MIN64_SGPR 1
#endif
__device__ __forceinline__ unsigned long long min_u64(unsigned long long a, unsigned long long b)
{
#if RVB_MIN64_SGPR
    unsigned long long mask;
    uint32_t lo, hi;
    // (s_nop 1: a VALU-written SGPR needs two wait states before a VALU reads it as a mask)
    asm("v_cmp_lt_u64_e64 %0, %3, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, %6, %5, %0\n\tv_cndmask_b32_e64 %2, %8, %7, %0"
        : "=&s"(mask), "=&v"(lo), "=&v"(hi)
        : "v"(a), "v"(b), "v"((uint32_t) a), "v"((uint32_t) b), "v"((uint32_t) (a >> 32)), "v"((uint32_t) (b >> 32)));
    return ((unsigned long long) hi << 32) | lo;
#else
    return a < b ? a : b;
#endif
}

// Population count of a wave mask as a 32-bit scalar (the builtin's 64-bit result drags the comparisons that follow
// onto the VALU as 64-bit compares).
__device__ __forceinline__ int scalar_popcount(unsigned long long mask)
{
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(mask) : "scc");
    return n;
}

// Closest-hit job loop with SCHEDULED step kinds over the wave's 16 quads (path_kernel, RVB_PATH_JOBS=2).
// A quad is in one of four states, all encoded in `ref`: at a node (bit 31 clear), at a leaf (bit 31 set), query
// finished (NONE), out of jobs (IDLE).  The while-while loop above runs node steps until the LAST quad has reached a
// leaf, so on incoherent rays (every bounce after the first) only ~7 of 16 quads do useful work in a node step.  Here
// a step kind is executed for the quads in that state while the others keep theirs.  Rounds 1-3 chose the kind by a
// majority vote per iteration (host replay on workload C2, tools/travsim.cpp: wave-level node steps per bounce 37 -> 28,
// quads active per node step 6.7 -> 8.9, wave instructions per bounce -13 %); round 4 replaced the vote by a fixed
// cycle with thresholds (node step, leaf step if a third of the live lanes wait for one, shading step if a quarter do:
// 27.6 + 4.8 + 2.4 -> 22.0 + 5.7 + 3.8 steps per 16 ray-bounces, tools/travforms.cpp) — see traverse_pairs_vote,
// "THE SCHEDULE", for the measurements.  (The function names keep the word `vote`.)
template <class Job>
__device__ __forceinline__ void traverse_jobs_vote(const SceneDev & sc, uint32_t * __restrict__ stack, Job & job,
                                                   lds_float4_ptr lds_nodes = nullptr)
{
    const uint32_t IDLE = 0xFFFFFFFEu;
    const uint32_t c = threadIdx.x & 3u;
    const uint32_t lane_base4 = (threadIdx.x & 60u) << 2;
    const uint32_t lane_bit = 1u << c, lt_mask = lane_bit - 1u;
    const char * node_base = reinterpret_cast<const char *>(sc.nodes);
    const uint32_t child_off = 16u * c;
    const float neg_cull = -sc.cull_abs, cull_scale = 1.0f + sc.cull_rel;
    v3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
    float tmax = 0.0f;
    const unsigned long long NO_HIT_KEY = (0x7F800000ull << 32) | NONE;
    const char * tri_base = reinterpret_cast<const char *>(sc.tris);      // wave-uniform base + 32-bit byte offset, like the nodes
    float ix = 0.0f, iy = 0.0f, iz = 0.0f, oix = 0.0f, oiy = 0.0f, oiz = 0.0f;
    unsigned long long best_key = NO_HIT_KEY;                              // (distance bits, triangle index) of the closest hit so far
    uint32_t sp = 0, ref = IDLE;
#if RVB_QUAD_SELECT
    uint32_t selx = 0, sely = 0, selz = 0;
#define RVB_QUAD_SEL_SET() selx = slab_selector(ix); sely = slab_selector(iy); selz = slab_selector(iz);
#define RVB_QUAD_SLAB(n, tn) slab_select(n, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, job.skip_ref(), tn)
#else
#define RVB_QUAD_SEL_SET()
#define RVB_QUAD_SLAB(n, tn) slab(n, ix, iy, iz, oix, oiy, oiz, limit, neg_cull, job.skip_ref(), tn)
#endif
#define RVB_RESET_QUERY()                                                         \
    {                                                                             \
        ix = clamp_inv(d.x); iy = clamp_inv(d.y); iz = clamp_inv(d.z);            \
        oix = o.x * ix; oiy = o.y * iy; oiz = o.z * iz;                           \
        RVB_QUAD_SEL_SET()                                                        \
        best_key = NO_HIT_KEY; sp = 0; ref = 0;                                   \
    }
    if (job.next(o, d, tmax)) RVB_RESET_QUERY()
    int n_active = 0;                    // lanes that carry a ray (not IDLE): changes in shading steps only
    auto leaf_step = [&]() {
        if ((int32_t) ref < (int32_t) IDLE) {
            const uint32_t first = ref & 0x0FFFFFFFu;
            const uint32_t count = ((ref >> 28) & 7u) + 1u;
            float dist = 0.0f;
            uint32_t idx = NONE;
            if (c < count) {
                const float4 * tp = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + c));
                float4 ta = tp[0], tb = tp[1], tc = tp[2];
                asm volatile("" : "+v"(ta.x), "+v"(tb.x), "+v"(tc.x));     // all three loads leave before the first use
                dist = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d);
                idx = __float_as_uint(tc.y);
            }
            // kernel.cpp:180-188 — smallest distance wins, equal distances go to the lower index.  A candidate
            // distance is > EPSILON > 0, and positive floats order like their bit patterns, so (distance, index)
            // is ONE unsigned 64-bit key: the quad minimum and the comparison with the best so far are three
            // 64-bit compares.  "No hit" is (+inf, NONE), the largest key a lane can hold.
            const bool valid = c < count && dist > RVB_EPSILON;
            unsigned long long key = valid ? (((unsigned long long) __float_as_uint(dist) << 32) | idx) : NO_HIT_KEY;
            key = min_u64(key, dpp_u64<QP_SWAP1>(key));
            key = min_u64(key, dpp_u64<QP_SWAP2>(key));
            best_key = min_u64(best_key, key);
            if (sp > 0) { --sp; ref = stack[sp * QUADS_PER_BLOCK]; } else ref = NONE;
        }
    };
    auto shading_step = [&]() {
        if (ref == NONE) {
            Hit h;
            h.t = __uint_as_float((uint32_t) (best_key >> 32));
            h.tri = (uint32_t) best_key;
            job.done(h.tri != NONE, h);
            ref = IDLE;
            if (job.next(o, d, tmax)) RVB_RESET_QUERY()
        }
        n_active = scalar_popcount(__builtin_amdgcn_ballot_w64(ref != IDLE));
    };
    auto node_step = [&]() {
        if ((int32_t) ref >= 0) {
#if RVB_LDS_NODES
            uint4 n;
            if (ref < RVB_LDS_NODES * 64u) { const nt_float4 t = lds_nodes[(ref | child_off) >> 4]; n = make_uint4(__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w)); }
            else n = *reinterpret_cast<const uint4 *>(node_base + (ref | child_off));
#else
            const uint4 n = *reinterpret_cast<const uint4 *>(node_base + (ref | child_off));
#endif
            const float limit = fmaf(__uint_as_float((uint32_t) (best_key >> 32)), cull_scale, sc.cull_abs);
            float tn;
            const bool ok = RVB_QUAD_SLAB(n, tn);
            const uint32_t cref = n.w;
            const uint32_t key = ok ? ((__float_as_uint(fmaxf(tn, 0.0f)) & ~3u) | c) : NONE;
            uint32_t kmin = min(key, dpp_u<QP_SWAP1>(key));
            kmin = min(kmin, dpp_u<QP_SWAP2>(kmin));
            if (kmin == NONE) {
                if (sp > 0) { --sp; ref = stack[sp * QUADS_PER_BLOCK]; } else ref = NONE;
            } else {
                const uint32_t winner = kmin & 3u;
                uint32_t okmask = ok ? lane_bit : 0u;
                okmask |= dpp_u<QP_SWAP1>(okmask);
                okmask |= dpp_u<QP_SWAP2>(okmask);
                const uint32_t rest = okmask & ~(1u << winner);
                if (ok && c != winner)
                    stack[(sp + __popc(rest & lt_mask)) * QUADS_PER_BLOCK] = cref;
                sp += __popc(rest);
                ref = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (lane_base4 + (winner << 2)), (int) cref);
            }
        }
    };
    n_active = scalar_popcount(__builtin_amdgcn_ballot_w64(ref != IDLE));
    // (the schedule of traverse_pairs_vote: no vote — node step, leaf step if a third of the live lanes wait for one, shading step if a quarter do)
    for (;;) {
        if (n_active == 0)
            break;
        bool ran = __builtin_amdgcn_ballot_w64((int32_t) ref >= 0) != 0ull;
        node_step();
        const int n_leaf = scalar_popcount(__builtin_amdgcn_ballot_w64((int32_t) ref < (int32_t) IDLE));   // signed: leaves are < -2
        if (n_leaf && (RVB_CYCLE_LEAF_NUM * n_leaf >= RVB_CYCLE_LEAF_DEN * n_active || !ran)) {
            leaf_step();
            ran = true;
        }
        const int n_done = scalar_popcount(__builtin_amdgcn_ballot_w64(ref == NONE));
        if (n_done && (RVB_CYCLE_DONE_NUM * n_done >= RVB_CYCLE_DONE_DEN * n_active || !ran))
            shading_step();
    }
#undef RVB_RESET_QUERY
}

// TWO LANES PER RAY (path_kernel at RVB_PATH_LANES = 2): a lane owns two children of a node and two triangles of a leaf, a wave
// carries 32 rays.  The vote, the stack handling, the reductions and the loads' addressing are per-RAY work that every lane of
// the ray repeats: with two lanes instead of four a node step costs ~1.45x the instructions for twice the rays.  (One lane per
// ray would be cheaper still per ray, but 100 k rays are then 1.5 waves per SIMD, too few to cover a node fetch.)
// stack: this pair's column of the LDS stack, entries PAIRS_PER_BLOCK words apart.
#define PAIRS_PER_BLOCK 32
#ifndef RVB_PAIR_SELECT
#define RVB_PAIR_SELECT 1          // bit 0: path_pair_kernel, bit 1: shadow_pair_kernel use slab_select (near / far plane by the direction's sign)
                                   // instead of slab (min / max).  Measured at C2: path pairs 3.92 -> 3.80 ms, shadow pairs 1.28 -> 1.34 ms
#endif
// The node step of traverse_pairs_vote is written for ISSUE COST (round 4; measured as the build flag RVB_PAIR_PUSH_COUNTS against the
// hit-mask form it replaced, like the short vote — RVB_PAIR_SHORT_VOTE — and the chained node step — RVB_PAIR_CHAIN — further down: the
// flags were folded in once the measurements were committed, tools/r04*_*.sh name them).  In the pipeline (traces of the
// next group beside the binning of this one) the SIMDs issue vector instructions three quarters of the time, and the node step is two
// thirds of the path kernel's instructions; tools/inst_probe.hip measures two classes of them on gfx950 — v_fma / v_add / v_mul_f32,
// v_mov, two-operand integer add / and / or / xor / right shift and v_bitop3 issue at the full rate, everything else (comparisons,
// selects, min / max, DPP, v_perm, v_fma_mix, three-operand integer forms) at 0.6 of it (profiles/r04b_inst_probe.log).  The step now:
//   - pushes from COUNTS: a lane keeps the children whose key is not the pair's minimum, the second lane's entries go on top of the
//     first lane's, so one two-bit count crosses the pair (one DPP move) instead of the four-bit hit mask and its population counts;
//   - keys of the UNCLAMPED entry distance, compared as signed integers (no max(t, 0) per child; tools/travforms.cpp replays the same
//     number of node visits), built with one v_bitop3_b32;
//   - the winner's reference as (mine | theirs) with 0 in the lane that does not own it;
//   - the culling distance is state (changes in leaf steps, five times rarer than node steps); the stack pointer is an LDS byte address.
// 75 -> 57 vector instructions, 118 -> 91 issue units per node step (tools/isa_mix.py); same visits, same records, same bytes.
// Measured (profiles/r04_push_counts_n1.txt): pipeline 4.47-4.50 -> 4.37-4.40 ms per impulse response with the kernel capped at 80
// VGPRs (RVB_PAIR_WAVES = 6); uncapped it takes 84, loses a wave per SIMD to the kernels beside it and the pipeline is 8 % SLOWER
// (4.82-4.87 ms) — the register count of the path kernel matters more than its instruction count.  Alone (one trace of 100 k rays,
// bound by the latency of its chains) the kernel takes 3.49 ms either way.
#define RVB_PAIR_SLAB(SEL, n, tn, skip) ((SEL) ? slab_select(n, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn) \
                                               : slab(n, ix, iy, iz, oix, oiy, oiz, limit, neg_cull, skip, tn))
template <class Job>
__device__ __forceinline__ void traverse_pairs_vote(const SceneDev & sc, uint32_t * __restrict__ stack, Job & job,
                                                    lds_float4_ptr lds_nodes = nullptr)
{
    const uint32_t IDLE = 0xFFFFFFFEu;
    const uint32_t h = threadIdx.x & 1u;
    uint32_t c0 = 2u * h, c1 = c0 + 1u;                                    // the children this lane owns
    asm volatile("" : "+v"(c0), "+v"(c1));                                 // lane constants that stay in their registers (else recomputed in every node step)
    const char * node_base = reinterpret_cast<const char *>(sc.nodes);
    const char * tri_base = reinterpret_cast<const char *>(sc.tris);
    uint32_t child_off = 32u * h;
    asm volatile("" : "+v"(child_off));
    uint32_t clear2 = ~3u;
    asm volatile("" : "+v"(clear2));
    const float neg_cull = -sc.cull_abs, cull_scale = 1.0f + sc.cull_rel;
    const unsigned long long NO_HIT_KEY = (0x7F800000ull << 32) | NONE;
    v3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
    float tmax = 0.0f;
    float ix = 0.0f, iy = 0.0f, iz = 0.0f, oix = 0.0f, oiy = 0.0f, oiz = 0.0f;
    unsigned long long best_key = NO_HIT_KEY;
    // the stack pointer is the LDS byte address of the pair's next free row (rows are PAIRS_PER_BLOCK words apart)
    const uint32_t PAIR_ROW = PAIRS_PER_BLOCK * (uint32_t) sizeof(uint32_t);
    const uint32_t bottom = (uint32_t) (uintptr_t) (lds_u32_ptr) stack;
    typedef uint32_t walk_t __attribute__((ext_vector_type(2)));
    walk_t walk = {IDLE, bottom};
#define ref walk.x
#define sp walk.y
#define RVB_PAIR_POP() { if (sp != bottom) { sp -= PAIR_ROW; ref = *(lds_u32_ptr) (uintptr_t) sp; } else ref = NONE; }
#define RVB_PAIR_EMPTY() sp = bottom
    uint32_t selx = 0, sely = 0, selz = 0;
    float limit = 0.0f;                  // culling distance of the best hit so far: changes in leaf steps, is read in node steps
#define RVB_PAIR_LIMIT() limit = fmaf(__uint_as_float((uint32_t) (best_key >> 32)), cull_scale, sc.cull_abs)
#define RVB_RESET_QUERY()                                                         \
    {                                                                             \
        ix = clamp_inv(d.x); iy = clamp_inv(d.y); iz = clamp_inv(d.z);            \
        oix = o.x * ix; oiy = o.y * iy; oiz = o.z * iz;                           \
        selx = slab_selector(ix); sely = slab_selector(iy); selz = slab_selector(iz); \
        best_key = NO_HIT_KEY; RVB_PAIR_EMPTY(); ref = 0; RVB_PAIR_LIMIT();       \
    }
#if RVB_STAMPS
    // diagnostic builds.  -DRVB_STAMPS=1: where a wave's cycles go — [0] vote, [1] node step until its two loads are back, [2] the rest of the
    // node step (incl. the wait for the popped entry), [3] / [4] the same for leaf steps, [5] shading steps; [6..8] step counts.  Any RVB_STAMPS
    // (2 = these alone, the loop runs at its own pace): [9] shader cycles and [11] 100-MHz ticks of the whole loop — their quotient is the
    // clock the chip holds under this load (MI355X_MICROARCH.md "DVFS give-back") —, [10] waves
    unsigned long long sv[6] = {0, 0, 0, 0, 0, 0}, sn[3] = {0, 0, 0}, t_loop, r_loop, t_a = 0, t_b = 0, t_c = 0;
    STAMP(t_loop)
    { __builtin_amdgcn_sched_barrier(0); r_loop = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
    t_c = t_loop;
#endif
    if (job.next(o, d, tmax)) RVB_RESET_QUERY()
    int n_active = 0;                    // lanes that carry a ray (not IDLE): changes in shading steps only
    // the three step kinds of the loop (inlined where the schedule below calls them)
    auto leaf_step = [&]() {
        RVB_MARK("leaf");
#if RVB_STAMPS == 1
        if ((int32_t) ref < (int32_t) IDLE) {
            const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
            const float4 * q0 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (h < count ? h : 0u)));
            const float4 * q1 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (h + 2u < count ? h + 2u : 0u)));
            float4 w0 = q0[0], w1 = q0[2], w2 = q1[0], w3 = q1[2];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0.x), "+v"(w1.x), "+v"(w2.x), "+v"(w3.x) :: "memory");
        }
        STAMP(t_b)
        sv[3] += t_b - t_a; sn[1] += 1;
#endif
        if ((int32_t) ref < (int32_t) IDLE) {
            // triangles h and h + 2 of the leaf (a two-triangle leaf gives each lane one)
            const uint32_t first = ref & 0x0FFFFFFFu;
            const uint32_t count = ((ref >> 28) & 7u) + 1u;
            const uint32_t j0 = h, j1 = h + 2u;
            const float4 * tp0 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (j0 < count ? j0 : 0u)));
            const float4 * tp1 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (j1 < count ? j1 : 0u)));
            float4 ta = tp0[0], tb = tp0[1], tc = tp0[2], ua = tp1[0], ub = tp1[1], uc = tp1[2];
            asm volatile("" : "+v"(ta.x), "+v"(tb.x), "+v"(tc.x), "+v"(ua.x), "+v"(ub.x), "+v"(uc.x));   // all six loads leave before the first use
            const float dist0 = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d);
            const float dist1 = mt_intersect(mk3(ua.x, ua.y, ua.z), mk3(ua.w, ub.x, ub.y), mk3(ub.z, ub.w, uc.x), o, d);
            // kernel.cpp:180-188 — smallest distance wins, equal distances go to the lower index: one unsigned 64-bit key
            const bool valid0 = j0 < count && dist0 > RVB_EPSILON, valid1 = j1 < count && dist1 > RVB_EPSILON;
            const unsigned long long k0 = valid0 ? (((unsigned long long) __float_as_uint(dist0) << 32) | __float_as_uint(tc.y)) : NO_HIT_KEY;
            const unsigned long long k1 = valid1 ? (((unsigned long long) __float_as_uint(dist1) << 32) | __float_as_uint(uc.y)) : NO_HIT_KEY;
            unsigned long long key = min_u64(k0, k1);
            key = min_u64(key, dpp_u64<QP_SWAP1>(key));
            best_key = min_u64(best_key, key);
            RVB_PAIR_LIMIT();
            RVB_PAIR_POP()
        }
#if RVB_STAMPS == 1
        STAMP(t_c)
        sv[4] += t_c - t_b;
#endif
    };
    auto shading_step = [&]() {
        RVB_MARK("done");
        if (ref == NONE) {
            Hit hit;
            hit.t = __uint_as_float((uint32_t) (best_key >> 32));
            hit.tri = (uint32_t) best_key;
            job.done(hit.tri != NONE, hit);
            ref = IDLE;
            if (job.next(o, d, tmax)) RVB_RESET_QUERY()
        }
        n_active = scalar_popcount(__builtin_amdgcn_ballot_w64(ref != IDLE));
#if RVB_STAMPS == 1
        STAMP(t_c)
        sv[5] += t_c - t_a; sn[2] += 1;
        t_a = t_c;
#endif
    };
    auto node_step = [&]() {
        RVB_MARK("node");
#if RVB_STAMPS == 1
        if ((int32_t) ref >= 0) {
            const uint4 * pp = reinterpret_cast<const uint4 *>(node_base + (ref | child_off));
            uint4 w0 = pp[0], w1 = pp[1];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0.x), "+v"(w1.x) :: "memory");     // the step's own loads hit the L1 afterwards
        }
        STAMP(t_b)
        sv[1] += t_b - t_a; sn[0] += 1;
#endif
        if ((int32_t) ref >= 0) {
#if RVB_LDS_NODES
            uint4 n0, n1;
            if (ref < RVB_LDS_NODES * 64u) {
                const nt_float4 t0 = lds_nodes[(ref | child_off) >> 4], t1 = lds_nodes[((ref | child_off) >> 4) + 1];
                n0 = make_uint4(__float_as_uint(t0.x), __float_as_uint(t0.y), __float_as_uint(t0.z), __float_as_uint(t0.w));
                n1 = make_uint4(__float_as_uint(t1.x), __float_as_uint(t1.y), __float_as_uint(t1.z), __float_as_uint(t1.w));
            } else {
                const uint4 * np = reinterpret_cast<const uint4 *>(node_base + (ref | child_off));
                n0 = np[0]; n1 = np[1];
            }
#else
            const uint4 * np = reinterpret_cast<const uint4 *>(node_base + (ref | child_off));
            const uint4 n0 = np[0], n1 = np[1];
#endif
            float tn0, tn1;
            const bool ok0 = RVB_PAIR_SLAB(RVB_PAIR_SELECT & 1, n0, tn0, job.skip_ref());
            const bool ok1 = RVB_PAIR_SLAB(RVB_PAIR_SELECT & 1, n1, tn1, job.skip_ref());
            // the hit children's keys: entry distance (its two low bits give way to the child number), compared as SIGNED integers —
            // negative distances (the origin is inside the box, or the box a rounding behind it) come before all others, in any
            // order; tools/travforms.cpp replays the same number of node visits as with keys of max(distance, 0)
            const uint32_t NO_CHILD = 0x7FFFFFFFu;
            // ((distance & ~3) | child) as one v_bitop3_b32 with register operands: issues at the rate of v_fma_f32, the and_or
            // form at 0.6 of it (profiles/r04b_inst_probe.log)
            const uint32_t key0 = ok0 ? __builtin_amdgcn_bitop3_b32(__float_as_uint(tn0), clear2, c0, 0xEA) : NO_CHILD;
            const uint32_t key1 = ok1 ? __builtin_amdgcn_bitop3_b32(__float_as_uint(tn1), clear2, c1, 0xEA) : NO_CHILD;
            uint32_t kmin = (uint32_t) min((int32_t) key0, (int32_t) key1);
            kmin = (uint32_t) min((int32_t) kmin, (int32_t) dpp_u<QP_SWAP1>(kmin));
            if (kmin == NO_CHILD) {
                RVB_PAIR_POP()
            } else {
                // the pair's pushes in child order (as below) from the lanes' COUNTS: a lane's kept children go on top of the other
                // lane's if it is the pair's second lane, so one 2-bit count crosses the pair instead of the hit mask, and a lane's
                // rows follow from its own two flags (the keys name the child: key == kmin is the winner)
                const bool other0 = key0 != kmin, other1 = key1 != kmin;
                const bool keep0 = ok0 && other0, keep1 = ok1 && other1;
                const uint32_t first = keep0 ? PAIR_ROW : 0u;                          // counts in bytes of stack rows
                const uint32_t n_mine = first + (keep1 ? PAIR_ROW : 0u);
                const uint32_t n_theirs = dpp_u<QP_SWAP1>(n_mine);
                const uint32_t row = __umul24(n_theirs, h) + sp;                       // sp + (h ? n_theirs : 0) as one v_mad_u32_u24
                if (keep0)
                    *(lds_u32_ptr) (uintptr_t) row = n0.w;
                if (keep1)
                    *(lds_u32_ptr) (uintptr_t) (row + first) = n1.w;
                // the winner is the child whose key IS kmin (keys carry the child number)
                const uint32_t mine = other1 ? (other0 ? 0u : n0.w) : n1.w;            // 0 in the lane that does not own it
                sp += n_mine + n_theirs;
                ref = mine | dpp_u<QP_SWAP1>(mine);
            }
        }
#if RVB_STAMPS == 1
        STAMP(t_c)
        sv[2] += t_c - t_b;
#endif
    };
    // THE SCHEDULE (round 4).  Rounds 1-3 voted: every iteration three ballots, and the step kind most lanes waited for was executed.  A wave's time,
    // though, goes into the LATENCY of its own instruction stream (tools/pair_stamps.py: 300 of an iteration's 1 900 cycles were the vote's dependent
    // scalar chain), so the vote was first shortened (one ballot while the lanes at a node are a majority: two-lane kernel alone 3.54 -> 3.46 ms,
    // four-lane kernel 3.54 -> 3.31), then a node step was chained behind every leaf and shading step (same steps, 28.7 votes instead of 36.2 per
    // 32 ray-bounces: 3.43 -> 3.27 / 3.32 -> 3.22 ms) — and then dropped: every iteration is a node step for the lanes at a node, then a leaf step if a
    // third of the live lanes wait for one, then a shading step if a quarter of them do (or if nothing else could run).  tools/travforms.cpp
    // (TRAVFORMS_CYCLE) replays 23.7 node + 6.3 leaf + 3.2 shading steps per 32 ray-bounces at C2 where the majority vote takes 28.6 + 5.2 + 2.6 (C4:
    // 26.5 + 5.9 + 3.3 against 32.0 + 4.9 + 2.6): lanes waiting at a leaf need not become the largest group before they are served, and the node steps
    // run fuller (0.61 of the lanes instead of 0.52).  Two-lane kernel alone 3.27 -> 3.16 ms (100 k rays), 1.86 -> 1.80 ms per 100 k rays at 800 k;
    // four-lane kernel 3.27 -> 3.13 ms; pipeline 4.22 -> 4.12 ms per IR (profiles/r04d_cycle*_n1.txt; thresholds of 25-40 % all within 1 %).
    // Same queries, same results: the schedule only decides WHEN a lane's next step runs.  What did not help a wave's latency: one dword of the next
    // node requested a step ahead (a third L1 access per step costs more than its head start: 3.40 -> 3.81 ms), two node steps per iteration.
    n_active = scalar_popcount(__builtin_amdgcn_ballot_w64(ref != IDLE));
    for (;;) {
        RVB_MARK("vote");
        if (n_active == 0)
            break;
        bool ran = __builtin_amdgcn_ballot_w64((int32_t) ref >= 0) != 0ull;
#if RVB_STAMPS == 1
        STAMP(t_a)
        sv[0] += t_a - t_c;
#endif
        node_step();
#if RVB_STAMPS == 1
        t_a = t_c;
#endif
        const int n_leaf = scalar_popcount(__builtin_amdgcn_ballot_w64((int32_t) ref < (int32_t) IDLE));
        if (n_leaf && (RVB_CYCLE_LEAF_NUM * n_leaf >= RVB_CYCLE_LEAF_DEN * n_active || !ran)) {
            leaf_step();
            ran = true;
#if RVB_STAMPS == 1
            t_a = t_c;
#endif
        }
        const int n_done = scalar_popcount(__builtin_amdgcn_ballot_w64(ref == NONE));
        if (n_done && (RVB_CYCLE_DONE_NUM * n_done >= RVB_CYCLE_DONE_DEN * n_active || !ran))
            shading_step();
        RVB_MARK("loop_end");
    }
#if RVB_STAMPS
    if (sc.stamps) {
        STAMP(t_b)
        unsigned long long r_end;
        { __builtin_amdgcn_sched_barrier(0); r_end = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
        if ((threadIdx.x & 63u) == 0) {
            for (int i = 0; i < 6; ++i) atomicAdd(sc.stamps + i, sv[i]);
            for (int i = 0; i < 3; ++i) atomicAdd(sc.stamps + 6 + i, sn[i]);
            atomicAdd(sc.stamps + 9, t_b - t_loop);
            atomicAdd(sc.stamps + 10, 1ull);
            atomicAdd(sc.stamps + 11, r_end - r_loop);
        }
    }
#endif
#undef RVB_RESET_QUERY
#undef RVB_PAIR_POP
#undef RVB_PAIR_EMPTY
#undef ref
#undef sp
#undef RVB_PAIR_LIMIT
}

// Any-hit query with two lanes per ray (shadow_pair_kernel): is there a triangle with EPSILON < distance <= tmax (the negation of
// reference kernel.cpp:295).  Lockstep like traverse_quad<true>: the 32 pairs of the wave start a query together and leave the
// loops as they finish; no visiting order (the lowest hit child is entered, the others pushed).
#define QP_PAIR_LO 0xA0   // quad_perm [0,0,2,2]: both lanes of a pair read its even lane
#define QP_PAIR_HI 0xF5   // quad_perm [1,1,3,3]: ... its odd lane
__device__ __forceinline__ bool traverse_pair_any(const SceneDev & sc, const v3 o, const v3 d, const float tmax,
                                                  uint32_t * __restrict__ stack, const uint32_t skip)
{
    const uint32_t h = threadIdx.x & 1u;
    const uint32_t c0 = 2u * h;
    const uint32_t bit0 = 1u << c0, bit1 = 2u << c0, lt0 = bit0 - 1u, lt1 = bit1 - 1u;
    const char * node_base = reinterpret_cast<const char *>(sc.nodes);
    const char * tri_base = reinterpret_cast<const char *>(sc.tris);
    const uint32_t child_off = 32u * h;                                    // (pinned in a register it would be the 81st: a wave per SIMD less)
    const float neg_cull = -sc.cull_abs;
    const float limit = fmaf(tmax, 1.0f + sc.cull_rel, sc.cull_abs);
    const float ix = clamp_inv(d.x), iy = clamp_inv(d.y), iz = clamp_inv(d.z);
    const float oix = o.x * ix, oiy = o.y * iy, oiz = o.z * iz;
    const uint32_t selx = slab_selector(ix), sely = slab_selector(iy), selz = slab_selector(iz);
    uint32_t sp = 0, ref = 0;
    for (;;) {
        while (!(ref & RVB_BVH_LEAF)) {
            const uint4 * np = reinterpret_cast<const uint4 *>(node_base + (ref | child_off));
            const uint4 n0 = np[0], n1 = np[1];
            float tn0, tn1;
            const bool ok0 = RVB_PAIR_SLAB(RVB_PAIR_SELECT & 2, n0, tn0, skip);
            const bool ok1 = RVB_PAIR_SLAB(RVB_PAIR_SELECT & 2, n1, tn1, skip);
            uint32_t okmask = (ok0 ? bit0 : 0u) | (ok1 ? bit1 : 0u);
            okmask |= dpp_u<QP_SWAP1>(okmask);
            if (okmask == 0u) {
                if (sp > 0) { --sp; ref = stack[sp * PAIRS_PER_BLOCK]; } else ref = NONE;
                continue;
            }
            const uint32_t rest = okmask & (okmask - 1u);             // all hit children but the lowest
            const uint32_t winner_bit = okmask ^ rest;
            if (ok0 && bit0 != winner_bit)
                stack[(sp + __popc(rest & lt0)) * PAIRS_PER_BLOCK] = n0.w;
            if (ok1 && bit1 != winner_bit)
                stack[(sp + __popc(rest & lt1)) * PAIRS_PER_BLOCK] = n1.w;
            sp += __popc(rest);
            const uint32_t mine = (winner_bit & 0xAu) ? n1.w : n0.w;  // children 1, 3 are the lanes' second child
            const uint32_t theirs = dpp_u<QP_SWAP1>(mine);
            ref = (winner_bit & (bit0 | bit1)) ? mine : theirs;
        }
        if (ref == NONE)
            return false;
        // triangles h and h + 2 of the leaf
        const uint32_t first = ref & 0x0FFFFFFFu;
        const uint32_t count = ((ref >> 28) & 7u) + 1u;
        const uint32_t j0 = h, j1 = h + 2u;
        const float4 * tp0 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (j0 < count ? j0 : 0u)));
        const float4 * tp1 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (j1 < count ? j1 : 0u)));
        float4 ta = tp0[0], tb = tp0[1], tc = tp0[2], ua = tp1[0], ub = tp1[1], uc = tp1[2];
        asm volatile("" : "+v"(ta.x), "+v"(tb.x), "+v"(tc.x), "+v"(ua.x), "+v"(ub.x), "+v"(uc.x));
        const float dist0 = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d);
        const float dist1 = mt_intersect(mk3(ua.x, ua.y, ua.z), mk3(ua.w, ub.x, ub.y), mk3(ub.z, ub.w, uc.x), o, d);
        uint32_t hit = ((j0 < count && dist0 > RVB_EPSILON && dist0 <= tmax) || (j1 < count && dist1 > RVB_EPSILON && dist1 <= tmax)) ? 1u : 0u;
        hit |= dpp_u<QP_SWAP1>(hit);
        if (hit)
            return true;
        if (sp > 0) { --sp; ref = stack[sp * PAIRS_PER_BLOCK]; } else return false;
    }
}

// A single query through the same loop (the quad's lanes return together).
struct OneShotJob {
    v3 o, d;
    float tmax;
    bool pending, hit;
    Hit result;
    uint32_t skip;
    __device__ __forceinline__ uint32_t skip_ref() const { return skip; }
    __device__ __forceinline__ bool next(v3 & o_, v3 & d_, float & tmax_)
    {
        if (!pending) return false;
        pending = false;
        o_ = o; d_ = d; tmax_ = tmax;
        return true;
    }
    __device__ __forceinline__ void done(bool h, const Hit & r) { hit = h; result = r; }
};

template <bool ANY>
__device__ __forceinline__ bool traverse_quad(const SceneDev & sc, const v3 o, const v3 d, const float tmax,
                                              uint32_t * __restrict__ stack, Hit & hit, const uint32_t skip = RVB_BVH_EMPTY)
{
    OneShotJob job = {o, d, tmax, true, false, {0.0f, NONE}, skip};
    traverse_jobs<ANY>(sc, stack, job);
    hit = job.result;
    return job.hit;
}

__device__ __forceinline__ v3 ld3(const float * p) { return mk3(p[0], p[1], p[2]); }

// ------------------------------------------------------------------------------------------------
// Work record left by path_kernel in impulses[ray*nrefl + bounce] (64 B; quad lane c stores chunk c):
//   chunk 0,1  newVol = -volume * specular                        (kernel.cpp:461)
//   chunk 2    intersection.xyz, DIFF = |dot(normal, dir)|         (kernel.cpp:459, :478)
//   chunk 3    newDist, own-plane threshold of the shadow ray, triangle index, pair + 1 = valid   (kernel.cpp:460)
// shadow_kernel turns it into the final Impulse in place.
// One ray's bounce chain as a Job: next() hands out the current ray, done() shades the hit
// (kernel.cpp:459-461, :478), stores the work record and reflects (kernel.cpp:492-501).
// Copies the scene's surface table (64 B per surface) behind the traversal stack in LDS when the launch reserved
// room for it (TraceArgs::lds_surfaces = number of surfaces staged, 0 = none).  Single-wave workgroups: the
// barrier is only the wait for the wave's own LDS writes.
__device__ __forceinline__ float4 lds_load4(lds_float4_ptr p, uint32_t i)
{
    const nt_float4 t = p[i];
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ lds_float4_ptr stage_surfaces(const TraceArgs & a, uint32_t * lds_after_stack)
{
    if (!a.lds_surfaces)
        return nullptr;
    float4 * dst = reinterpret_cast<float4 *>(lds_after_stack);
    const float4 * src = reinterpret_cast<const float4 *>(a.scene.surfaces);
    for (uint32_t i = threadIdx.x; i < 4u * a.lds_surfaces; i += WAVE)
        dst[i] = src[i];
    __syncthreads();
    return (lds_float4_ptr) dst;
}

// LANES = 4: quad lane c stores chunk c.  LANES = 2: lane c of the pair stores chunks c and c + 2.
// COLD (the two-lane kernel): what only the shading step touches — the lane's four band volumes and the path length so far — lives
// in LDS between bounces ([5][64] words behind the surface table, one column per lane) instead of in five registers that the
// compiler would otherwise keep through every node and leaf step: with them the kernel fits the 72-register budget of seven
// waves per SIMD without scratch spills (the compiler's own choice at that budget spills two values the LEAF step reloads).
// Measured at workload C2 (round 3, profiles/r03_pair_cold_state_n1.txt): with COLD the group kernel needs 72 registers and no
// scratch, all 6 250 waves of two 100 k-ray traces are resident at once (80 registers: 6 144 slots, and the 106 waves that start
// when the first ones finish run a whole 128-bounce chain almost alone: 196 608 rays 4.97 ms, 200 000 rays 6.15 ms), and the launch
// takes 5.36 instead of 6.15 ms.  In the bench pipeline it LOSES (4.81 against 4.59 ms per IR): the other group's sort, shadow and
// binning kernels used to run in that long thin tail and now stretch by what the path kernel gained (chain per context 16.7 against
// 16.6 ms).  Off by default; -DRVB_PAIR_COLD=1 for callers whose path launches run alone.
#ifndef RVB_PAIR_COLD
#define RVB_PAIR_COLD 0
#endif
#define RVB_KEY_RUN 32u      // grouping keys per run: 32 x 2 bytes = one 64-byte segment
__device__ __forceinline__ uint32_t lane_id_here()
{
    uint32_t lane;      // (volatile: recomputed where it is used instead of being kept in a register across the traversal loop)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}
template <bool SURF_LDS, int LANES = 4, bool COLD = false>
struct PathJob {
    const TraceArgs & a;
    uint32_t ray;                        // < 2^32 / 9 (rvb_trace checks)
    uint32_t c;
    v3 o, d;
    float distance;                      // (COLD: in LDS)
    float4 vol;                          // lane 0 (and 2): bands 0-3, lane 1 (and 3): bands 4-7 — the chunk the lane stores (COLD: in LDS)
    uint32_t index;
    bool alive;
    lds_float4_ptr surf_lds;             // the surface table staged in LDS (stage_surfaces); unused when !SURF_LDS
    uint32_t pair_tag;                   // (source, microphone) pair of this ray + 1: what marks its work records as valid
    uint32_t skip;                       // own-plane subtree of the triangle the current segment starts on (TriShade, bvh.h)
    bool unit;                           // the ray's direction has unit length (the own-plane rule is derived for |d| = 1)
    float * cold;                        // COLD: this workgroup's [5][64] words in LDS
    uint16_t * key_rows;                 // key runs (TraceArgs::sort_keys16): this workgroup's [rays][RVB_KEY_RUN] 16-bit keys in LDS

    // Record-grouping keys as 64-byte RUNS.  The grouping key of a record (the leaf position of the triangle hit, 16 significant
    // bits) used to leave as one 4-byte store per record, 8 KB apart in a ray's row: 51 MB of keys cost 0.4 GB of HBM writes (a
    // partial-line write each; WRITE_SIZE 1.29 GB per launch for 0.87 GB of records and keys).  Now lane 1 of the ray parks the
    // 16-bit key in LDS and, every RVB_KEY_RUN bounces, the ray's lanes write the run as whole 16-byte pieces of one 64-byte segment.
    __device__ __forceinline__ uint16_t * key_row() const { return key_rows + (lane_id_here() >> (LANES == 4 ? 2 : 1)) * RVB_KEY_RUN; }
    __device__ __forceinline__ void flush_key_run(const uint16_t * row, uint64_t first_record) const
    {
        const uint4 * src = reinterpret_cast<const uint4 *>(row);
        uint4 * dst = reinterpret_cast<uint4 *>(a.sort_keys16 + first_record);
        if (LANES == 4) {
            dst[c] = src[c];
        } else {
            dst[2 * c] = src[2 * c];
            dst[2 * c + 1] = src[2 * c + 1];
        }
    }

    __device__ __forceinline__ uint32_t skip_ref() const { return skip; }
    __device__ __forceinline__ bool next(v3 & o_, v3 & d_, float & tmax)
    {
        if (!alive || index >= a.nreflections)
            return false;
        o_ = o;
        d_ = d;
        tmax = 0.0f;
        return true;
    }
    __device__ __forceinline__ void done(bool hit, const Hit & h)
    {
        if (!hit) {                                                  // kernel.cpp:372-375
            alive = false;
            return;
        }
        const float4 * shade = reinterpret_cast<const float4 *>(a.scene.shade + h.tri);       // 32 B: normal + surface, own-plane skip
        const float4 sh = shade[0], sk = shade[1];
        const v3 normal = mk3(sh.x, sh.y, sh.z);
        const uint32_t surface = __float_as_uint(sh.w);
        // the specular row hangs off a dependent load (triangle -> surface -> row): from LDS it costs ~64 cycles
        // instead of another L2 round trip; each lane reads the half row of the four bands it carries
        const uint32_t half = c & 1u;
        float4 sp;
        if (SURF_LDS) sp = lds_load4(surf_lds, 4 * surface + half);
        else sp = reinterpret_cast<const float4 *>(a.scene.surfaces + surface)[half];
        float * mine = nullptr;
        if (COLD) {
            mine = cold + lane_id_here();
            vol = make_float4(mine[0], mine[64], mine[128], mine[192]);
            distance = mine[256];
        }
        const v3 p = o + d * h.t;                                    // kernel.cpp:459
        const float new_dist = distance + h.t;                       // kernel.cpp:460
        vol = make_float4(-vol.x * sp.x, -vol.y * sp.y, -vol.z * sp.z, -vol.w * sp.w);   // kernel.cpp:461
        const float diff = fabsf(dot3(normal, d));                   // kernel.cpp:478
        // Own-plane skip (bvh.h): the rays that START at p — the reflected ray and the shadow ray — may pass over the subtree of
        // this triangle's plane patch when they leave the plane steeply enough: |cos| > skip_a + skip_b * (segment length).  The
        // reflected ray's |cos| is `diff` (reflection keeps it); the shadow kernel compares its own against the threshold the
        // record carries (+inf: no skip).
        const float threshold = unit ? fmaf(sk.z, h.t, sk.y) : __builtin_inff();
        skip = diff > threshold ? __float_as_uint(sk.x) : RVB_BVH_EMPTY;
        float4 chunk = vol;
        const float4 tail = make_float4(new_dist, threshold, __uint_as_float(h.tri), __uint_as_float(pair_tag));   // tag: pair + 1, non-zero = valid
        if (LANES == 4) {
            if (c == 2) chunk = make_float4(p.x, p.y, p.z, diff);
            else if (c == 3) chunk = tail;
        }
        // (the product is formed here, one v_mad_u64_u32 per bounce: hoisted out of the loop it would hold two more VGPRs for
        // the whole traversal, which at the 64-register budget of 8 waves per SIMD means a spill)
        uint32_t ray_here = ray;
        asm volatile("" : "+v"(ray_here));
        const uint64_t record = (uint64_t) ray_here * a.nreflections + index;
#if RVB_PROBE_NO_STORES          // diagnostic build (wrong output): what the record stores cost the bounce chain
        if (new_dist == 1.2345f) {
#endif
        store_stream(reinterpret_cast<float4 *>(a.impulses + record) + c, chunk);
        if (LANES == 2)
            store_stream(reinterpret_cast<float4 *>(a.impulses + record) + c + 2, c == 0 ? make_float4(p.x, p.y, p.z, diff) : tail);
        if (c == 0 && index < RVB_NUM_IMAGE_SOURCE - 1)
            a.early[ray * (RVB_NUM_IMAGE_SOURCE - 1) + index] = h.tri;
        if (a.sort_keys16) {                                         // (wave-uniform) keys leave in runs, see flush_key_run
            uint16_t * row = key_row();
            const uint32_t at = index & (RVB_KEY_RUN - 1u);
            if (c == 1) row[at] = (uint16_t) (__float_as_uint(sk.w) >> a.key_shift);      // the triangle's position in leaf order (rvb_set_scene put it there)
            if (at == RVB_KEY_RUN - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // lane 1's 16-bit store before the 16-byte loads of the ray's other lane(s)
                flush_key_run(row, record - at);
            }
        } else if (c == 1 && a.sort_keys) {
            a.sort_keys[record] = __float_as_uint(sk.w);
        }
#if RVB_PROBE_NO_STORES
        }
#endif
        if (COLD) {
            mine[0] = vol.x; mine[64] = vol.y; mine[128] = vol.z; mine[192] = vol.w;
            mine[256] = new_dist;
        }
        d = reflect3(normal, d);                                     // kernel.cpp:492-499
        o = p;
        if (!COLD) distance = new_dist;
        ++index;
    }
};

// After the traversal: an escaped ray leaves its remaining slots zero-filled (reference rayverb.cpp:600-603 zero-fills the whole
// buffer before every launch; here only the few slots that need it are written) and their grouping keys "no record".
template <class Job, int LANES>
__device__ __forceinline__ void finish_escaped_ray(const TraceArgs & a, Job & job, const uint64_t ray)
{
    if (job.index >= a.nreflections)
        return;
    for (uint32_t i = job.index; i < a.nreflections; ++i) {
        const uint64_t record = ray * a.nreflections + i;
        store_stream(reinterpret_cast<float4 *>(a.impulses + record) + job.c, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        if (LANES == 2)
            store_stream(reinterpret_cast<float4 *>(a.impulses + record) + job.c + 2, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        if (job.c == 1 && a.sort_keys)
            a.sort_keys[record] = NONE;
    }
    if (a.sort_keys16) {
        // the run the ray was in: its remaining keys become "no record", then it leaves like any other; whole runs after it directly
        uint16_t * row = job.key_row();
        uint32_t i = job.index;
        const uint32_t at = i & (RVB_KEY_RUN - 1u);
        if (at) {
            if (job.c == 1)
                for (uint32_t k = at; k < RVB_KEY_RUN; ++k) row[k] = 0xFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            job.flush_key_run(row, ray * a.nreflections + (i - at));
            i += RVB_KEY_RUN - at;
        }
        const uint4 none = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        for (; i < a.nreflections; i += RVB_KEY_RUN) {
            uint4 * dst = reinterpret_cast<uint4 *>(a.sort_keys16 + ray * a.nreflections + i);
            if (LANES == 4) dst[job.c] = none;
            else { dst[2 * job.c] = none; dst[2 * job.c + 1] = none; }
        }
    }
}

// 64 VGPRs = 8 waves per SIMD: one resident round holds 8 x 1024 x 16 = 131 072 rays, so the 125 k rays per GPU of
// workload C3 still run as one round (at 72 VGPRs / 7 waves they took 5.1 ms instead of 4.3 ms).
// WAVES = 7 (72 VGPRs) is the build for launches that fit in seven waves per SIMD (<= 114 688 rays: the 100 k rays of workload C2 are
// 6.1 waves per SIMD): with the registers of slab_select and no spill, 3.66 -> 3.50 ms at C2; the 8-wave build keeps larger launches
// (up to 131 072 rays) in one resident round (3.66 -> 3.58 ms at C2 with slab_select and two spilled registers).
template <bool SURF_LDS, int WAVES>
__global__ __launch_bounds__(WAVE, WAVES) void path_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][QUADS_PER_BLOCK]
    const uint32_t q = threadIdx.x >> 2;
    const uint64_t ray = (uint64_t) blockIdx.x * QUADS_PER_BLOCK + q;
    const lds_float4_ptr surf_lds = stage_surfaces(a, stack_lds + a.stack_entries * QUADS_PER_BLOCK);
#if RVB_LDS_NODES
    lds_float4_ptr lds_nodes;
    {
        uint4 * dst = reinterpret_cast<uint4 *>(stack_lds + a.stack_entries * QUADS_PER_BLOCK) + 4u * a.lds_surfaces;
        const uint4 * src = reinterpret_cast<const uint4 *>(a.scene.nodes);
        const uint32_t count = 4u * (RVB_LDS_NODES < a.scene_nodes ? RVB_LDS_NODES : a.scene_nodes);
        for (uint32_t i = threadIdx.x; i < 4u * RVB_LDS_NODES; i += WAVE)
            dst[i] = i < count ? src[i] : make_uint4(0xFC007C00u, 0xFC007C00u, 0xFC007C00u, RVB_BVH_EMPTY);
        __syncthreads();
        lds_nodes = (lds_float4_ptr) dst;
    }
#endif
    if (ray >= a.nrays)
        return;                                   // whole quads leave together
    uint32_t pair = 0, local = (uint32_t) ray;
    v3 source = ld3(a.source);
    if (a.npairs > 1) {                           // wave-uniform: several (source, microphone) pairs share the launch
        pair = (uint32_t) ray / a.rays_per_pair;
        local = (uint32_t) ray - pair * a.rays_per_pair;
        const float4 s4 = a.pair_sources[pair];
        source = mk3(s4.x, s4.y, s4.z);
    }
    const float4 d4 = a.directions[local];
    const float len2 = d4.x * d4.x + d4.y * d4.y + d4.z * d4.z;
    PathJob<SURF_LDS> job = {a, (uint32_t) ray, threadIdx.x & 3u, source, mk3(d4.x, d4.y, d4.z), 0.0f,
                   make_float4(1.0f, 1.0f, 1.0f, 1.0f), 0u, true, surf_lds, pair + 1u, RVB_BVH_EMPTY, fabsf(len2 - 1.0f) < 1e-3f, nullptr,
                   reinterpret_cast<uint16_t *>(stack_lds + a.stack_entries * QUADS_PER_BLOCK + 16u * a.lds_surfaces + (RVB_LDS_NODES * 16u))};
#if RVB_PATH_JOBS == 2
#if RVB_LDS_NODES
    traverse_jobs_vote(a.scene, stack_lds + q, job, lds_nodes);
#else
    traverse_jobs_vote(a.scene, stack_lds + q, job);
#endif
#elif RVB_PATH_JOBS
    traverse_jobs<false>(a.scene, stack_lds + q, job);
#else
    v3 o, d;
    float tmax;
    while (job.next(o, d, tmax)) {
        Hit h;
        const bool hit = traverse_quad<false>(a.scene, o, d, tmax, stack_lds + q, h, job.skip);
        job.done(hit, h);
    }
#endif
    finish_escaped_ray<PathJob<SURF_LDS>, 4>(a, job, ray);
    if (job.c == 0)
        atomicAdd(a.executed, (unsigned long long) job.index);
}

// path_kernel with two lanes per ray (traverse_pairs_vote): 32 rays per single-wave workgroup.
#ifndef RVB_PAIR_WAVES
#define RVB_PAIR_WAVES 6            // register cap: 80 VGPRs, so that six waves fit a SIMD beside the other kernels' (see the node step of traverse_pairs_vote);
                                    // 7 (72 VGPRs) spills ten registers: pipeline 4.52-4.54 ms against 4.37-4.40, and 4.70 against 4.47 when LDS
                                    // allows the seventh wave too (no key runs: profiles/r04c_occupancy_n1.txt); 8 (64 VGPRs): 5.9 ms
#endif
template <bool SURF_LDS>
__device__ __forceinline__ void path_pair_body(const TraceArgs & a, const uint32_t block)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][PAIRS_PER_BLOCK]
#if RVB_PATH_PRIO
    __builtin_amdgcn_s_setprio(RVB_PATH_PRIO);
#endif
    const uint32_t q = threadIdx.x >> 1;
    const uint64_t ray = (uint64_t) block * PAIRS_PER_BLOCK + q;
    const lds_float4_ptr surf_lds = stage_surfaces(a, stack_lds + a.stack_entries * PAIRS_PER_BLOCK);
#if RVB_LDS_NODES
    lds_float4_ptr lds_nodes;
    {
        uint4 * dst = reinterpret_cast<uint4 *>(stack_lds + a.stack_entries * PAIRS_PER_BLOCK) + 4u * a.lds_surfaces;
        const uint4 * src = reinterpret_cast<const uint4 *>(a.scene.nodes);
        const uint32_t count = 4u * (RVB_LDS_NODES < a.scene_nodes ? RVB_LDS_NODES : a.scene_nodes);
        for (uint32_t i = threadIdx.x; i < 4u * RVB_LDS_NODES; i += WAVE)
            dst[i] = i < count ? src[i] : make_uint4(0xFC007C00u, 0xFC007C00u, 0xFC007C00u, RVB_BVH_EMPTY);
        __syncthreads();
        lds_nodes = (lds_float4_ptr) dst;
    }
#endif
    if (ray >= a.nrays)
        return;                                   // whole pairs leave together
    uint32_t pair = 0, local = (uint32_t) ray;
    v3 source = ld3(a.source);
    if (a.npairs > 1) {
        pair = (uint32_t) ray / a.rays_per_pair;
        local = (uint32_t) ray - pair * a.rays_per_pair;
        const float4 s4 = a.pair_sources[pair];
        source = mk3(s4.x, s4.y, s4.z);
    }
    const float4 d4 = a.directions[local];
    const float len2 = d4.x * d4.x + d4.y * d4.y + d4.z * d4.z;
    // the cold words sit behind the stack and the surface table (rvb_pair_lds_bytes)
    float * cold = reinterpret_cast<float *>(stack_lds + a.stack_entries * PAIRS_PER_BLOCK + 16u * a.lds_surfaces + (RVB_LDS_NODES * 16u));
    if (RVB_PAIR_COLD) {
        cold[threadIdx.x] = 1.0f; cold[64 + threadIdx.x] = 1.0f; cold[128 + threadIdx.x] = 1.0f; cold[192 + threadIdx.x] = 1.0f;
        cold[256 + threadIdx.x] = 0.0f;
    }
    PathJob<SURF_LDS, 2, RVB_PAIR_COLD != 0> job = {a, (uint32_t) ray, threadIdx.x & 1u, source, mk3(d4.x, d4.y, d4.z), 0.0f,
                   make_float4(1.0f, 1.0f, 1.0f, 1.0f), 0u, true, surf_lds, pair + 1u, RVB_BVH_EMPTY, fabsf(len2 - 1.0f) < 1e-3f, cold,
                   reinterpret_cast<uint16_t *>(cold + (RVB_PAIR_COLD ? 5u * WAVE : 0u))};
#if RVB_LDS_NODES
    traverse_pairs_vote(a.scene, stack_lds + q, job, lds_nodes);
#else
    traverse_pairs_vote(a.scene, stack_lds + q, job);
#endif
    finish_escaped_ray<PathJob<SURF_LDS, 2, RVB_PAIR_COLD != 0>, 2>(a, job, ray);
    if (job.c == 0)
        atomicAdd(a.executed, (unsigned long long) job.index);
}

// Several traces (contexts: their own rays, buffers, source and microphone) in ONE launch.  Two path kernels launched side by side
// are not scheduled alike — the first one's waves are older and issue first, the second runs on alone at half the occupancy (4.9 and
// 7.9 ms at workload C2) — whereas the waves of one launch advance together.  first_block[k] = first workgroup of trace k.
struct TraceGroup {
    uint32_t count;
    uint32_t first_block[RVB_MAX_GROUP + 1];
    TraceArgs trace[RVB_MAX_GROUP];
};
template <bool SURF_LDS>
__global__ __launch_bounds__(WAVE, RVB_PAIR_WAVES) void path_pair_group_kernel(TraceGroup g)
{
    uint32_t which = 0;
    for (uint32_t k = 1; k < g.count; ++k)
        which += blockIdx.x >= g.first_block[k] ? 1u : 0u;
    path_pair_body<SURF_LDS>(g.trace[which], blockIdx.x - g.first_block[which]);
}

// ONE LANE PER RAY (path_lane_group_kernel, round 4): 64 rays per single-wave workgroup, every lane walks its own ray — it tests the
// four children of its node and the up-to-four triangles of its leaf itself, nothing is exchanged between lanes, and the stack is the
// lane's own column in LDS.  What the lanes of a pair (or quad) repeat per ray — the vote's state, the stack pointer, the child
// keys, the lane exchange, the 64-bit key reductions — is paid once per ray, and a wave step serves 64 rays: tools/travforms.cpp
// replays C2 at 29.3 node + 5.4 leaf + 2.7 shading wave steps per 64 ray-bounces (pairs: 28.5 + 5.2 + 2.6 per 32), i.e. a quarter to
// a third fewer wave instructions per ray-bounce with the step costs of this kernel's ISA.  The price is half the waves again (100 k
// rays are 1.5 waves per SIMD) and longer steps, so a launch is bound by the latency of one wave's chain unless about 400 k rays are
// in flight: rvb_path_lanes_for picks it for group launches of that size only.  Same arithmetic, same records, same bytes as the
// other two path kernels (tests/test_gpu_parity.py runs every trace case with all three).
#define LANE_RAYS 64
typedef __attribute__((address_space(3))) void * lds_void_ptr;
typedef const __attribute__((address_space(1))) void * global_void_ptr;
#ifndef RVB_LANE_WAVES
#define RVB_LANE_WAVES 4            // waves per SIMD the register budget allows (128 VGPRs)
#endif
// Node fetch of the one-lane kernel.  A lane that reads the 64 bytes of ITS node with four 16-byte loads makes four L1 accesses per
// node visit (the texture addresser coalesces the lanes of ONE instruction, not the instructions of one lane): PMC at 800 k rays —
// 87 L1 accesses per ray-bounce against 50 (pairs) and 33 (quads), TA stalled by the L1 a quarter of its busy time, and the kernel
// slower than the pairs although it issues 27 % fewer vector instructions (profiles/r04_path_pmc_by_lanes_800k_rays_n1.txt).
// RVB_LANE_COOP = 1: the four lanes of a QUAD fetch each other's nodes — in load s every lane of the quad reads "its" child (16 bytes)
// of the node of the quad's lane s, one contiguous 64-byte access per quad like the quad kernel's — straight into LDS
// (global_load_lds_dwordx4: no registers in between), and each lane then reads its own node's four children back with four
// ds_read_b128.  Load s of the wave lands at stage + s * RVB_LANE_STAGE_STRIDE + lane * 16 (the extra 16 bytes per load put the four
// reads of a quad's lanes on different banks).
// MEASURED, and slower still (profiles/r04_rays_sweep_by_lanes_n1.txt): 800 k rays 3.02 ms per 100 k rays against 2.35 with the plain
// loads (pairs 2.07), 100 k rays 4.48 against 3.94 — the detour through LDS adds a dependent round trip to every node step and its 4 KB
// of LDS per wave cost a fifth of the occupancy.  Off; the shipped kernels stay the pair / quad kernels.
#ifndef RVB_LANE_COOP
#define RVB_LANE_COOP 0
#endif
#define RVB_LANE_STAGE_STRIDE 1040u
template <bool SURF_LDS>
__device__ __forceinline__ void path_lane_body(const TraceArgs & a, const uint32_t block)
{
    // LDS of the workgroup: [stack_entries + 1][64] stack words (a lane's column; one slack row: pushes store first and advance if
    // kept), the surface table, [64][RVB_KEY_RUN] 16-bit grouping keys (rvb_lane_lds_bytes)
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];
    const uint32_t IDLE = 0xFFFFFFFEu;
    const uint32_t lane = threadIdx.x;
    const uint64_t ray = (uint64_t) block * LANE_RAYS + lane;
    uint32_t * const after_stack = stack_lds + (a.stack_entries + 1u) * LANE_RAYS;
    const lds_float4_ptr surf_lds = stage_surfaces(a, after_stack);
    uint16_t * const key_row = reinterpret_cast<uint16_t *>(after_stack + 16u * a.lds_surfaces) + lane * RVB_KEY_RUN;
    // (behind the key runs when there are any: rvb_lane_lds_bytes) the landing area of the cooperative node fetch
#if RVB_LANE_COOP
    uint32_t * const stage = after_stack + 16u * a.lds_surfaces + (a.sort_keys16 ? LANE_RAYS * RVB_KEY_RUN / 2u : 0u);
#endif
    const bool in_range = ray < a.nrays;
#if !RVB_LANE_COOP
    if (!in_range)
        return;
#endif
    // (with the cooperative fetch the lanes behind the last ray stay: they carry no ray, but fetch for their quad's other lanes)
    uint32_t pair = 0, local = in_range ? (uint32_t) ray : 0u;
    v3 o = ld3(a.source);
    if (a.npairs > 1) {
        pair = (uint32_t) ray / a.rays_per_pair;
        local = (uint32_t) ray - pair * a.rays_per_pair;
        const float4 s4 = a.pair_sources[pair];
        o = mk3(s4.x, s4.y, s4.z);
    }
    const float4 d4 = a.directions[local];
    v3 d = mk3(d4.x, d4.y, d4.z);
    const bool unit = fabsf(d4.x * d4.x + d4.y * d4.y + d4.z * d4.z - 1.0f) < 1e-3f;
    const uint32_t pair_tag = pair + 1u;
    float4 vol_lo = make_float4(1.0f, 1.0f, 1.0f, 1.0f), vol_hi = vol_lo;      // kernel.cpp:322-323
    float distance = 0.0f;
    uint32_t index = 0, skip = RVB_BVH_EMPTY;

    const char * node_base = reinterpret_cast<const char *>(a.scene.nodes);
    const char * tri_base = reinterpret_cast<const char *>(a.scene.tris);
    const float neg_cull = -a.scene.cull_abs, cull_scale = 1.0f + a.scene.cull_rel;
    const unsigned long long NO_HIT_KEY = (0x7F800000ull << 32) | NONE;
    const lds_u32_ptr bottom = (lds_u32_ptr) stack_lds + lane;
    lds_u32_ptr sp = bottom;
    float ix = 0.0f, iy = 0.0f, iz = 0.0f, oix = 0.0f, oiy = 0.0f, oiz = 0.0f;
    uint32_t selx = 0, sely = 0, selz = 0;
    unsigned long long best_key = NO_HIT_KEY;
    uint32_t ref = IDLE;
#define RVB_RESET_QUERY()                                                         \
    {                                                                             \
        ix = clamp_inv(d.x); iy = clamp_inv(d.y); iz = clamp_inv(d.z);            \
        oix = o.x * ix; oiy = o.y * iy; oiz = o.z * iz;                           \
        selx = slab_selector(ix); sely = slab_selector(iy); selz = slab_selector(iz); \
        best_key = NO_HIT_KEY; sp = bottom; ref = 0;                              \
    }
    if (in_range && index < a.nreflections) RVB_RESET_QUERY()
    for (;;) {
        RVB_MARK("vote");
        const unsigned long long m_node = __builtin_amdgcn_ballot_w64((int32_t) ref >= 0);
        const unsigned long long m_done = __builtin_amdgcn_ballot_w64(ref == NONE);
        const unsigned long long m_leaf = __builtin_amdgcn_ballot_w64((int32_t) ref < (int32_t) IDLE);
        const int n_node = scalar_popcount(m_node), n_done = scalar_popcount(m_done), n_leaf = scalar_popcount(m_leaf);
        if ((n_node | n_done | n_leaf) == 0)
            break;
        if (n_node >= n_leaf && n_node >= n_done) {
            RVB_MARK("node");
#if RVB_LANE_COOP
            {
                // every lane of the wave is here (the branch is wave-uniform): load s fetches the node of each quad's lane s, if that lane
                // is at a node (the condition is quad-uniform: whole quads skip the load)
                const uint32_t child = 16u * (lane & 3u);
#define RVB_COOP_LOAD(S)                                                                                                         \
                {                                                                                                                \
                    const uint32_t ref_s = quad_bcast_u<S>(ref);                                                                 \
                    if ((int32_t) ref_s >= 0)                                                                                    \
                        __builtin_amdgcn_global_load_lds((global_void_ptr) (node_base + (ref_s | child)),                        \
                                                         (lds_void_ptr) (stage + S * (RVB_LANE_STAGE_STRIDE / 4u)), 16, 0, 0);   \
                }
                RVB_COOP_LOAD(0) RVB_COOP_LOAD(1) RVB_COOP_LOAD(2) RVB_COOP_LOAD(3)
#undef RVB_COOP_LOAD
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the LDS-DMA loads are not in the compiler's books
            }
#endif
            if ((int32_t) ref >= 0) {
#if RVB_LANE_COOP
                // this lane's node: load (lane & 3) of the wave, this quad's 64 bytes
                const uint4 * np = reinterpret_cast<const uint4 *>(stage + (lane & 3u) * (RVB_LANE_STAGE_STRIDE / 4u) + (lane & ~3u) * 4u);
#else
                const uint4 * np = reinterpret_cast<const uint4 *>(node_base + ref);
#endif
                const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                const float limit = fmaf(__uint_as_float((uint32_t) (best_key >> 32)), cull_scale, a.scene.cull_abs);
                float tn0, tn1, tn2, tn3;
                const bool ok0 = slab_select(n0, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn0);
                const bool ok1 = slab_select(n1, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn1);
                const bool ok2 = slab_select(n2, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn2);
                const bool ok3 = slab_select(n3, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn3);
                // nearest hit child first, by the same (entry distance | child) keys as the other path kernels: the visiting order, and
                // with it the culling, is theirs
                const uint32_t key0 = ok0 ? ((__float_as_uint(fmaxf(tn0, 0.0f)) & ~3u) | 0u) : NONE;
                const uint32_t key1 = ok1 ? ((__float_as_uint(fmaxf(tn1, 0.0f)) & ~3u) | 1u) : NONE;
                const uint32_t key2 = ok2 ? ((__float_as_uint(fmaxf(tn2, 0.0f)) & ~3u) | 2u) : NONE;
                const uint32_t key3 = ok3 ? ((__float_as_uint(fmaxf(tn3, 0.0f)) & ~3u) | 3u) : NONE;
                const uint32_t kmin = min(min(key0, key1), min(key2, key3));
                // the other hit children go on the stack in child order: store, then advance past the store if it is kept
                *sp = n0.w; sp += (ok0 && key0 != kmin) ? LANE_RAYS : 0;
                *sp = n1.w; sp += (ok1 && key1 != kmin) ? LANE_RAYS : 0;
                *sp = n2.w; sp += (ok2 && key2 != kmin) ? LANE_RAYS : 0;
                *sp = n3.w; sp += (ok3 && key3 != kmin) ? LANE_RAYS : 0;
                if (kmin == NONE) {
                    if (sp != bottom) { sp -= LANE_RAYS; ref = *sp; } else ref = NONE;
                } else {
                    const uint32_t lo = (kmin & 1u) ? n1.w : n0.w, hi = (kmin & 1u) ? n3.w : n2.w;
                    ref = (kmin & 2u) ? hi : lo;
                }
            }
        } else if (n_leaf >= n_done) {
            RVB_MARK("leaf");
            if ((int32_t) ref < (int32_t) IDLE) {
                const uint32_t first = ref & 0x0FFFFFFFu;
                const uint32_t count = ((ref >> 28) & 7u) + 1u;
                const float4 * tp0 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first));
                const float4 * tp1 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (1u < count ? 1u : 0u)));
                const float4 * tp2 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (2u < count ? 2u : 0u)));
                const float4 * tp3 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (3u < count ? 3u : 0u)));
                float4 ta = tp0[0], tb = tp0[1], tc = tp0[2], ua = tp1[0], ub = tp1[1], uc = tp1[2];
                float4 va = tp2[0], vb = tp2[1], vc = tp2[2], wa = tp3[0], wb = tp3[1], wc = tp3[2];
                // all twelve loads leave before the first use (one round trip per leaf step, not four)
                asm volatile("" : "+v"(ta.x), "+v"(tb.x), "+v"(tc.x), "+v"(ua.x), "+v"(ub.x), "+v"(uc.x),
                                  "+v"(va.x), "+v"(vb.x), "+v"(vc.x), "+v"(wa.x), "+v"(wb.x), "+v"(wc.x));
                const float dist0 = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d);
                const float dist1 = mt_intersect(mk3(ua.x, ua.y, ua.z), mk3(ua.w, ub.x, ub.y), mk3(ub.z, ub.w, uc.x), o, d);
                const float dist2 = mt_intersect(mk3(va.x, va.y, va.z), mk3(va.w, vb.x, vb.y), mk3(vb.z, vb.w, vc.x), o, d);
                const float dist3 = mt_intersect(mk3(wa.x, wa.y, wa.z), mk3(wa.w, wb.x, wb.y), mk3(wb.z, wb.w, wc.x), o, d);
                // kernel.cpp:180-188 — smallest distance wins, equal distances go to the lower index: one unsigned 64-bit key
                const bool valid0 = dist0 > RVB_EPSILON, valid1 = 1u < count && dist1 > RVB_EPSILON;
                const bool valid2 = 2u < count && dist2 > RVB_EPSILON, valid3 = 3u < count && dist3 > RVB_EPSILON;
                const unsigned long long k0 = valid0 ? (((unsigned long long) __float_as_uint(dist0) << 32) | __float_as_uint(tc.y)) : NO_HIT_KEY;
                const unsigned long long k1 = valid1 ? (((unsigned long long) __float_as_uint(dist1) << 32) | __float_as_uint(uc.y)) : NO_HIT_KEY;
                const unsigned long long k2 = valid2 ? (((unsigned long long) __float_as_uint(dist2) << 32) | __float_as_uint(vc.y)) : NO_HIT_KEY;
                const unsigned long long k3 = valid3 ? (((unsigned long long) __float_as_uint(dist3) << 32) | __float_as_uint(wc.y)) : NO_HIT_KEY;
                best_key = min_u64(min_u64(best_key, min_u64(k0, k1)), min_u64(k2, k3));
                if (sp != bottom) { sp -= LANE_RAYS; ref = *sp; } else ref = NONE;
            }
        } else {
            RVB_MARK("done");
            if (ref == NONE) {
                const uint32_t tri = (uint32_t) best_key;
                ref = IDLE;
                if (tri != NONE) {                                           // (else: the ray escaped, kernel.cpp:372-375)
                    const float t = __uint_as_float((uint32_t) (best_key >> 32));
                    // PathJob::done with one lane: the same operations on the same operands
                    const float4 * shade = reinterpret_cast<const float4 *>(a.scene.shade + tri);
                    const float4 sh = shade[0], sk = shade[1];
                    const v3 normal = mk3(sh.x, sh.y, sh.z);
                    const uint32_t surface = __float_as_uint(sh.w);
                    float4 s_lo, s_hi;
                    if (SURF_LDS) { s_lo = lds_load4(surf_lds, 4 * surface); s_hi = lds_load4(surf_lds, 4 * surface + 1); }
                    else { const float4 * row = reinterpret_cast<const float4 *>(a.scene.surfaces + surface); s_lo = row[0]; s_hi = row[1]; }
                    const v3 p = o + d * t;                                  // kernel.cpp:459
                    const float new_dist = distance + t;                     // kernel.cpp:460
                    vol_lo = make_float4(-vol_lo.x * s_lo.x, -vol_lo.y * s_lo.y, -vol_lo.z * s_lo.z, -vol_lo.w * s_lo.w);   // kernel.cpp:461
                    vol_hi = make_float4(-vol_hi.x * s_hi.x, -vol_hi.y * s_hi.y, -vol_hi.z * s_hi.z, -vol_hi.w * s_hi.w);
                    const float diff = fabsf(dot3(normal, d));               // kernel.cpp:478
                    const float threshold = unit ? fmaf(sk.z, t, sk.y) : __builtin_inff();      // own-plane skip (bvh.h)
                    skip = diff > threshold ? __float_as_uint(sk.x) : RVB_BVH_EMPTY;
                    const uint64_t record = (uint64_t) (uint32_t) ray * a.nreflections + index;
                    float4 * rec = reinterpret_cast<float4 *>(a.impulses + record);
                    store_stream(rec + 0, vol_lo);
                    store_stream(rec + 1, vol_hi);
                    store_stream(rec + 2, make_float4(p.x, p.y, p.z, diff));
                    store_stream(rec + 3, make_float4(new_dist, threshold, __uint_as_float(tri), __uint_as_float(pair_tag)));
                    if (index < RVB_NUM_IMAGE_SOURCE - 1)
                        a.early[(uint32_t) ray * (RVB_NUM_IMAGE_SOURCE - 1) + index] = tri;
                    if (a.sort_keys16) {                                     // (wave-uniform) grouping keys leave in 64-byte runs
                        const uint32_t at = index & (RVB_KEY_RUN - 1u);
                        key_row[at] = (uint16_t) (__float_as_uint(sk.w) >> a.key_shift);
                        if (at == RVB_KEY_RUN - 1u) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the row's 16-bit stores before its 16-byte loads
                            const uint4 * src = reinterpret_cast<const uint4 *>(key_row);
                            uint4 * dst = reinterpret_cast<uint4 *>(a.sort_keys16 + (record - at));
                            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                        }
                    } else if (a.sort_keys) {
                        a.sort_keys[record] = __float_as_uint(sk.w);
                    }
                    d = reflect3(normal, d);                                 // kernel.cpp:492-499
                    o = p;
                    distance = new_dist;
                    ++index;
                    if (index < a.nreflections) RVB_RESET_QUERY()
                }
            }
        }
        RVB_MARK("loop_end");
    }
#undef RVB_RESET_QUERY
    if (!in_range)
        return;
    // an escaped ray leaves its remaining slots zero-filled and their grouping keys "no record" (finish_escaped_ray)
    if (index < a.nreflections) {
        const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t i = index; i < a.nreflections; ++i) {
            float4 * rec = reinterpret_cast<float4 *>(a.impulses + ((uint64_t) (uint32_t) ray * a.nreflections + i));
            store_stream(rec + 0, zero); store_stream(rec + 1, zero); store_stream(rec + 2, zero); store_stream(rec + 3, zero);
            if (a.sort_keys)
                a.sort_keys[(uint64_t) (uint32_t) ray * a.nreflections + i] = NONE;
        }
        if (a.sort_keys16) {
            uint32_t i = index;
            const uint32_t at = i & (RVB_KEY_RUN - 1u);
            const uint4 none = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (at) {
                for (uint32_t k = at; k < RVB_KEY_RUN; ++k) key_row[k] = 0xFFFFu;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const uint4 * src = reinterpret_cast<const uint4 *>(key_row);
                uint4 * dst = reinterpret_cast<uint4 *>(a.sort_keys16 + ((uint64_t) (uint32_t) ray * a.nreflections + (i - at)));
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                i += RVB_KEY_RUN - at;
            }
            for (; i < a.nreflections; i += RVB_KEY_RUN) {
                uint4 * dst = reinterpret_cast<uint4 *>(a.sort_keys16 + ((uint64_t) (uint32_t) ray * a.nreflections + i));
                dst[0] = none; dst[1] = none; dst[2] = none; dst[3] = none;
            }
        }
    }
    atomicAdd(a.executed, (unsigned long long) index);
}

template <bool SURF_LDS>
__global__ __launch_bounds__(WAVE, RVB_LANE_WAVES) void path_lane_group_kernel(TraceGroup g)
{
    uint32_t which = 0;
    for (uint32_t k = 1; k < g.count; ++k)
        which += blockIdx.x >= g.first_block[k] ? 1u : 0u;
    path_lane_body<SURF_LDS>(g.trace[which], blockIdx.x - g.first_block[which]);
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

// A mirror plane of the image-source chain: the unit normal of a (mirrored) triangle and its first vertex.
// mirror_point (rvb_math.h, kernel.cpp:216-221) recomputes that normal — a cross product, a square root and three divisions —
// for every point it mirrors; here it is computed ONCE per plane with the same operations on the same operands, so the
// mirrored points are bit-identical.
struct MirrorPlane { v3 n, v0; };
__device__ __forceinline__ MirrorPlane mirror_plane(const TriVerts & t)
{
    MirrorPlane m;
    m.n = verts_normal(t);
    m.v0 = t.v0;
    return m;
}
__device__ __forceinline__ void mirror_point_on(v3 & p, const MirrorPlane & m)
{
    const float d = dot3(m.n, p - m.v0);
    p = p + ((-m.n) * d) * 2.0f;
}

// Image-source validation (kernel.cpp:379-457) in two kernels.
//
// A (ray, bounce) pair yields an image source iff (1) the ray from the source to the mirrored microphone crosses every mirrored
// triangle of the chain (Möller–Trumbore on the image triangles: arithmetic only), (2) each segment of the un-mirrored path is the
// closest hit of the real scene within +-EPSILON per component (a closest-hit query per segment), and (3) the last point sees the
// microphone (an any-hit query).  Hardly any pair passes (1) — a few hundred of 900 000 at workload C2 — and round 2's kernel (one lane
// per ray doing everything) took as long as its unluckiest LANE needed for up to eleven one-lane traversals in a row: 0.35 ms at 0.19
// lane use.  Now:
//   image_plan_kernel   one lane per ray walks its first nine bounces as before — the chain of mirrored triangles grown bounce by
//                       bounce, one mirror plane per bounce — but only evaluates (1) and appends the pairs that pass to a list;
//   image_check_kernel  FOUR lanes per listed pair and QUERY: rebuilds the pair's chain (all four lanes alike) and runs one of the
//                       queries of (2) and (3) with the quad traversal of the path kernel (four children / triangles per step instead
//                       of one); the pair's last query writes the image impulse.  The direct path (slot 0, one per source /
//                       microphone pair) is one more list entry.
// The operations on every value are the same as before, in the same order: results are bit-identical (tests/test_gpu_parity.py goldens).
#define RVB_IMAGE_DIRECT 0xFFFFFFFFu

// the ray's pair geometry (several (source, microphone) pairs may share a launch)
__device__ __forceinline__ void image_pair_of(const TraceArgs & a, uint32_t ray, uint32_t & pair, v3 & mic, v3 & source)
{
    pair = 0;
    mic = ld3(a.mic);
    source = ld3(a.source);
    if (a.npairs > 1) {
        pair = ray / a.rays_per_pair;
        const float4 m4 = a.pair_mics[pair], s4 = a.pair_sources[pair];
        mic = mk3(m4.x, m4.y, m4.z);
        source = mk3(s4.x, s4.y, s4.z);
    }
}

// One step of the chain (kernel.cpp:381-394): bounce `index`'s triangle through the planes so far, then the microphone through it.
struct ImageChain {
    TriVerts prev[RVB_NUM_IMAGE_SOURCE - 1];
    MirrorPlane plane[RVB_NUM_IMAGE_SOURCE - 1];
    v3 mic_reflection;
    __device__ __forceinline__ void extend(const SceneDev & sc, uint32_t index, uint32_t tri_here)
    {
        TriVerts current = load_corners(sc, tri_here);
        for (uint32_t j = 0; j < index; ++j) {
            mirror_point_on(current.v0, plane[j]);
            mirror_point_on(current.v1, plane[j]);
            mirror_point_on(current.v2, plane[j]);
        }
        prev[index] = current;
        plane[index] = mirror_plane(current);
        mirror_point_on(mic_reflection, plane[index]);
    }
    // the k-th crossing of the image ray, un-mirrored (kernel.cpp:406-414); false: the image ray misses image triangle k
    __device__ __forceinline__ bool crossing(uint32_t k, const v3 source, const v3 dir, v3 & ip) const
    {
        const float to_intersection = mt_intersect_verts(prev[k], source, dir);
        if (to_intersection <= RVB_EPSILON)
            return false;
        ip = source + dir * to_intersection;
        for (int l = (int) k - 1; l != -1; --l)
            mirror_point_on(ip, plane[l]);
        return true;
    }
};

__global__ __launch_bounds__(WAVE) void image_plan_kernel(TraceArgs a)
{
    const uint64_t ray = (uint64_t) blockIdx.x * WAVE + threadIdx.x;
    if (a.npairs <= 1 && ray == 0) {              // slot 0, the direct path (defined even for an empty ray set)
        const uint32_t at = atomicAdd(a.image_item_count, 1u);
        a.image_items[at] = ImageItem{0u, RVB_IMAGE_DIRECT};
    }
    if (ray >= a.nrays)
        return;
    uint32_t pair;
    v3 mic, source;
    image_pair_of(a, (uint32_t) ray, pair, mic, source);
    if (a.npairs > 1 && (uint32_t) ray == pair * a.rays_per_pair) {       // ... once per pair of a multi-pair launch
        const uint32_t at = atomicAdd(a.image_item_count, 1u);
        a.image_items[at] = ImageItem{pair, RVB_IMAGE_DIRECT};
    }
    const uint32_t per_ray = RVB_NUM_IMAGE_SOURCE - 1;
    const uint32_t * early = a.early + ray * per_ray;
    const uint32_t last = a.nreflections < per_ray ? a.nreflections : per_ray;
    ImageChain chain;
    chain.mic_reflection = mic;
    for (uint32_t index = 0; index < last; ++index) {
        const uint32_t tri_here = early[index];
        if (tri_here == NONE)
            break;                                // the ray escaped before this bounce
        chain.extend(a.scene, index, tri_here);
        const v3 dir = normalize3(chain.mic_reflection - source);      // kernel.cpp:396
        bool crosses = true;
        for (uint32_t k = 0; k != index + 1 && crosses; ++k) {
            v3 ip;
            crosses = chain.crossing(k, source, dir, ip);
        }
        if (crosses) {
            const uint32_t at = atomicAdd(a.image_item_count, 1u);     // (at most nrays * 9 + npairs entries: the list's capacity)
            a.image_items[at] = ImageItem{(uint32_t) ray, index};
            a.image_state[at] = 0u;
        }
    }
}

// reference kernel.cpp:274-296 (point_intersection) by the quad's four lanes
__device__ __forceinline__ bool point_visible_quad(const SceneDev & sc, v3 begin, v3 point, uint32_t * stack)
{
    const v3 b2p = point - begin;
    const float mag = length3(b2p);
    Hit h;
    return !traverse_quad<true>(sc, begin, normalize3(b2p), mag, stack, h);
}

// Four lanes per (listed pair, query): a pair at bounce `index` owns index + 1 closest-hit queries and one any-hit query, which do not
// depend on each other's results (the points come from the mirror chain, not from the queries), so they run side by side in
// RVB_IMAGE_QUERIES quad slots per pair instead of one after the other (one quad walking a ninth-bounce pair's ten queries took as long
// as round 2's whole kernel).  Every query adds its verdict to the pair's state word — low half: queries done, high half: queries
// failed — and the one whose add completes the pair writes the image impulse if none failed.
#define RVB_IMAGE_QUERIES (RVB_NUM_IMAGE_SOURCE + 1)
__global__ __launch_bounds__(WAVE) void image_check_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][QUADS_PER_BLOCK]
    const uint32_t q = threadIdx.x >> 2, c = threadIdx.x & 3u;
    uint32_t * stack = stack_lds + q;
    const uint64_t slots = (uint64_t) *a.image_item_count * RVB_IMAGE_QUERIES;
    for (uint64_t slot = (uint64_t) blockIdx.x * QUADS_PER_BLOCK + q; slot < slots; slot += (uint64_t) gridDim.x * QUADS_PER_BLOCK) {
        const uint32_t item = (uint32_t) (slot / RVB_IMAGE_QUERIES), k = (uint32_t) (slot % RVB_IMAGE_QUERIES);
        const ImageItem it = a.image_items[item];
        uint32_t pair;
        v3 mic, source;
        if (it.index == RVB_IMAGE_DIRECT) {
            if (k != 0)
                continue;
            // slot 0 (kernel.cpp:335-357): identical for every ray of the pair, computed once
            image_pair_of(a, it.ray * a.rays_per_pair, pair, mic, source);
            rvb_impulse direct;
            for (int b = 0; b < 8; ++b) direct.volume[b] = 0.0f;
            for (int b = 0; b < 4; ++b) direct.position[b] = 0.0f;
            direct.time = 0.0f;
            direct.pad_[0] = direct.pad_[1] = direct.pad_[2] = 0.0f;
            const bool visible = point_visible_quad(a.scene, source, mic, stack);
            if (c == 0) {
                if (visible) {
                    float one[8] = {1, 1, 1, 1, 1, 1, 1, 1};
                    make_image(a, mic, mic, source, one, direct);
                }
                a.direct[it.ray] = direct;
            }
            continue;
        }
        if (k > it.index + 1)
            continue;                             // this pair has fewer queries than slots
        image_pair_of(a, it.ray, pair, mic, source);
        const uint32_t * early = a.early + (uint64_t) it.ray * (RVB_NUM_IMAGE_SOURCE - 1);
        ImageChain chain;
        chain.mic_reflection = mic;
        for (uint32_t index = 0; index <= it.index; ++index)
            chain.extend(a.scene, index, early[index]);
        // kernel.cpp:396-440: query k < index + 1 is the k-th segment of the un-mirrored path, query index + 1 the view of the microphone
        const v3 dir = normalize3(chain.mic_reflection - source);
        v3 begin = source, ip = source;
        bool ok = true;
        if (k > 0) ok = chain.crossing(k - 1, source, dir, begin);        // (cannot fail: image_plan_kernel evaluated the same expression)
        if (k <= it.index) {
            ok = ok && chain.crossing(k, source, dir, ip);
            const v3 idir = normalize3(ip - begin);
            Hit h;
            const bool found = traverse_quad<false>(a.scene, begin, idir, 0.0f, stack, h);
            const float hd = found ? h.t : 0.0f;                          // Intersection {0, 0, false}
            const v3 nip = begin + idir * hd;
            const bool lo = (nip.x - RVB_EPSILON < ip.x) && (nip.y - RVB_EPSILON < ip.y) && (nip.z - RVB_EPSILON < ip.z);
            const bool hi = (ip.x < nip.x + RVB_EPSILON) && (ip.y < nip.y + RVB_EPSILON) && (ip.z < nip.z + RVB_EPSILON);
            ok = ok && found && lo && hi;
        } else {
            ok = point_visible_quad(a.scene, begin, mic, stack) && ok;    // kernel.cpp:431-440
        }
        if (c != 0)
            continue;
        const uint32_t before = atomicAdd(a.image_state + item, ok ? 1u : 0x10001u);
        if ((before & 0xFFFFu) + 1u != it.index + 2u || (before >> 16) != 0u || !ok)
            continue;                             // not the pair's last query, or one of them failed
        // kernel.cpp:442-456: the ray's volume BEFORE this bounce's surface is applied
        float volume[8];
        if (it.index == 0) {
            for (int b = 0; b < 8; ++b) volume[b] = 1.0f;
        } else {
            const float4 * rec = reinterpret_cast<const float4 *>(a.impulses + (uint64_t) it.ray * a.nreflections + (it.index - 1));
            const float4 v0 = rec[0], v1 = rec[1];
            volume[0] = v0.x; volume[1] = v0.y; volume[2] = v0.z; volume[3] = v0.w;
            volume[4] = v1.x; volume[5] = v1.y; volume[6] = v1.z; volume[7] = v1.w;
        }
        rvb_image_candidate cand;
        cand.ray = a.ray_offset + it.ray;
        cand.slot = it.index + 1;
        cand.index = early[it.index] + 1;
        make_image(a, mic, chain.mic_reflection, source, volume, cand.impulse);
        const uint32_t at = atomicAdd(a.candidate_count, 1u);
        a.candidates[at] = cand;
    }
}

// The shadow rays as Jobs: a quad walks the work records g, g + stride, ...; next() loads a record
// (the quad reads its 64 bytes as one line, lane c = chunk c) and aims at the microphone
// (kernel.cpp:463-469), done() finishes the Impulse in place (kernel.cpp:471-490).
template <bool SURF_LDS>
struct ShadowJob {
    const TraceArgs & a;
    uint32_t c;
    uint64_t g, stride, total;
    v3 mic;
    float airA, airB;                    // bands 2c, 2c+1: every lane evaluates two of the eight attenuations
    float4 * rec;
    float4 mine;
    v3 p;
    float diff, new_dist, mag;
    uint32_t surface;
    float tmin, tmax_seen;               // arrival-time range of the non-zero impulses this lane's quad produced
    uint32_t pair;                       // pair of the current record (several pairs per launch only)
    lds_float4_ptr surf_lds;             // surface table in LDS; unused when !SURF_LDS
    uint32_t skip;                       // own-plane subtree of the triangle the shadow ray starts on, RVB_BVH_EMPTY = none

    __device__ __forceinline__ uint32_t skip_ref() const { return skip; }
    __device__ __forceinline__ bool next(v3 & o_, v3 & d_, float & tmax)
    {
        while (g < total) {
            // with a.sort_order the quads of a wave take consecutive records of one bucket: shadow rays that
            // start within one triangle and all aim at the microphone walk the same BVH nodes
            rec = reinterpret_cast<float4 *>(a.impulses + (a.sort_order ? (uint64_t) a.sort_order[g] : g));
            g += stride;
            mine = load_stream(rec + c);
            // chunk 3 = (newDist, surface, triangle, valid); chunk 2 = (intersection, DIFF)
            const uint32_t tag = quad_bcast_u<3>(__float_as_uint(mine.w));
            if (tag == 0u)
                continue;                         // ray had already escaped: slot keeps its zero fill
            if (a.npairs > 1) {                   // the record's pair: its microphone, its time range
                const float4 m4 = a.pair_mics[tag - 1u];
                mic = mk3(m4.x, m4.y, m4.z);
                pair = tag - 1u;
            }
            new_dist = quad_bcast_f<3>(mine.x);
            const float threshold = quad_bcast_f<3>(mine.y);
            // the triangle's shading record (the quad's lanes read the same 32 bytes): surface, and the own-plane skip
            const float4 * shade = reinterpret_cast<const float4 *>(a.scene.shade + quad_bcast_u<3>(__float_as_uint(mine.z)));
            const float4 sh = shade[0];
            const uint32_t skip_ref = __float_as_uint(shade[1].x);
            surface = __float_as_uint(sh.w);
            p = mk3(quad_bcast_f<2>(mine.x), quad_bcast_f<2>(mine.y), quad_bcast_f<2>(mine.z));
            diff = quad_bcast_f<2>(mine.w);
            const v3 b2p = mic - p;               // kernel.cpp:282-286
            mag = length3(b2p);
            o_ = p;
            d_ = normalize3(b2p);
            tmax = mag;
            skip = fabsf(dot3(mk3(sh.x, sh.y, sh.z), d_)) > threshold ? skip_ref : RVB_BVH_EMPTY;
            return true;
        }
        return false;
    }
    __device__ __forceinline__ void done(bool blocked, const Hit &)
    {
        const bool visible = !blocked;
        const float dist = visible ? new_dist + mag : 0.0f;          // kernel.cpp:471
        float4 o = make_float4(0, 0, 0, 0);
        // attenuation of bands 2c, 2c+1 in this lane; lanes 0/1 then collect bands 0-3 / 4-7 by DPP
        float eA = 0.0f, eB = 0.0f;
        if (visible) {
            eA = air_attenuation(dist, airA) * 1.0f;
            eB = air_attenuation(dist, airB) * 1.0f;
        }
        const float e0 = dpp_f<0xE8>(eA), e1 = dpp_f<0xE8>(eB);     // quad_perm [0,2,2,3]: lane 0 <- 0, lane 1 <- 2
        const float e2 = dpp_f<0xED>(eA), e3 = dpp_f<0xED>(eB);     // quad_perm [1,3,2,3]: lane 0 <- 1, lane 1 <- 3
        if (c < 2) {
            if (visible) {
                float4 dc;                                           // diffuse coefficients of this lane's four bands
                if (SURF_LDS) dc = lds_load4(surf_lds, 4 * surface + 2 + c);
                else dc = reinterpret_cast<const float4 *>(a.scene.surfaces + surface)[2 + c];
                // kernel.cpp:480-485: newVol * attenuation * diffuse * DIFF, left to right
                o.x = ((mine.x * e0) * dc.x) * diff;
                o.y = ((mine.y * e1) * dc.y) * diff;
                o.z = ((mine.z * e2) * dc.z) * diff;
                o.w = ((mine.w * e3) * dc.w) * diff;
            }
        } else if (c == 2) {
            o = make_float4(p.x, p.y, p.z, 0.0f);
        } else {
            o.x = seconds_per_meter() * dist;                        // kernel.cpp:489
        }
        store_stream(rec + c, o);
        // inputs of findPredelay / MAX_SAMPLE (rayverb.h:49-74, rayverb.cpp:54-57) for free: an impulse
        // takes part iff any band is non-zero (kernel.cpp:524)
        const bool nonzero = quad_any(c < 2 && (o.x != 0.0f || o.y != 0.0f || o.z != 0.0f || o.w != 0.0f));
        if (nonzero) {
            const float t = seconds_per_meter() * dist;
            if (a.npairs > 1) {
                // one range per pair: a quad's lane 0 updates it, and only when a plain read says the value can still move
                if (c == 0) {
                    const volatile uint32_t * seen = a.time_range + 2u * pair;
                    if (t != 0.0f && __float_as_uint(t) < seen[0]) atomicMin(a.time_range + 2u * pair, __float_as_uint(t));
                    if (__float_as_uint(t) > seen[1]) atomicMax(a.time_range + 2u * pair + 1u, __float_as_uint(t));
                }
            } else {
                if (t != 0.0f) tmin = fminf(tmin, t);
                tmax_seen = fmaxf(tmax_seen, t);
            }
        }
    }
};

#ifndef RVB_SHADOW_PAIR_WAVES
#define RVB_SHADOW_PAIR_WAVES 5
#endif
#ifndef RVB_SHADOW_PRIO
#define RVB_SHADOW_PRIO 0
#endif
#ifndef RVB_PATH_PRIO
#define RVB_PATH_PRIO 0
#endif
#ifndef RVB_SHADOW_WAVES
#define RVB_SHADOW_WAVES 8     // 64 VGPRs (8 waves/SIMD): 1.845 -> 1.807 ms against 7
#endif
template <bool SURF_LDS>
__global__ __launch_bounds__(WAVE, RVB_SHADOW_WAVES) void shadow_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][QUADS_PER_BLOCK]
    const uint32_t c = threadIdx.x & 3u;
    const uint32_t q = threadIdx.x >> 2;
    ShadowJob<SURF_LDS> job = {a};
    job.c = c;
    job.g = (uint64_t) blockIdx.x * QUADS_PER_BLOCK + q;
    job.stride = (uint64_t) gridDim.x * QUADS_PER_BLOCK;
    job.total = a.nrays * (uint64_t) a.nreflections;
    job.mic = ld3(a.mic);
    job.pair = 0;
    job.airA = a.air[2 * c];
    job.airB = a.air[2 * c + 1];
    job.tmin = __builtin_inff();
    job.tmax_seen = 0.0f;
    job.surf_lds = stage_surfaces(a, stack_lds + a.stack_entries * QUADS_PER_BLOCK);
    job.skip = RVB_BVH_EMPTY;
#if RVB_SHADOW_JOBS
    traverse_jobs<true>(a.scene, stack_lds + q, job);
#else
    // one record per quad per pass: the 16 quads of the wave start and finish a pass together
    v3 o, d;
    float tmax;
    while (job.next(o, d, tmax)) {
        Hit h;
        const bool blocked = traverse_quad<true>(a.scene, o, d, tmax, stack_lds + q, h, job.skip);
        job.done(blocked, h);
    }
#endif
    float tmin = job.tmin, tmax_seen = job.tmax_seen;
    for (int off = 32; off > 0; off >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, off));
        tmax_seen = fmaxf(tmax_seen, __shfl_xor(tmax_seen, off));
    }
    if (threadIdx.x == 0 && a.npairs <= 1) {      // non-negative floats order like their bit patterns
        const volatile uint32_t * seen = a.time_range;    // skip the atomic when it cannot move the result (stale reads are harmless)
        if (tmin != __builtin_inff() && __float_as_uint(tmin) < seen[0]) atomicMin(a.time_range + 0, __float_as_uint(tmin));
        if (__float_as_uint(tmax_seen) > seen[1]) atomicMax(a.time_range + 1, __float_as_uint(tmax_seen));
    }
}

// shadow_kernel with two lanes per record: lane 0 carries chunks 0 and 2 of the 64-byte record (bands 0-3; hit point, DIFF), lane 1
// chunks 1 and 3 (bands 4-7; distance, own-plane threshold, triangle, tag).  Each lane evaluates the four attenuations of its bands.
template <bool SURF_LDS>
__global__ __launch_bounds__(WAVE, RVB_SHADOW_PAIR_WAVES) void shadow_pair_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries][PAIRS_PER_BLOCK]
#if RVB_SHADOW_PRIO
    __builtin_amdgcn_s_setprio(RVB_SHADOW_PRIO);      // experiment: the shadow waves win the issue arbitration against resident path waves of the other group
#endif
    const uint32_t h = threadIdx.x & 1u;
    const uint32_t q = threadIdx.x >> 1;
    uint32_t * stack = stack_lds + q;
    const lds_float4_ptr surf_lds = stage_surfaces(a, stack_lds + a.stack_entries * PAIRS_PER_BLOCK);
    const uint64_t stride = (uint64_t) gridDim.x * PAIRS_PER_BLOCK, total = a.nrays * (uint64_t) a.nreflections;
    v3 mic = ld3(a.mic);
    const float air0 = a.air[4 * h], air1 = a.air[4 * h + 1], air2 = a.air[4 * h + 2], air3 = a.air[4 * h + 3];
    float tmin = __builtin_inff(), tmax_seen = 0.0f;
    for (uint64_t g = (uint64_t) blockIdx.x * PAIRS_PER_BLOCK + q; g < total; g += stride) {
        float4 * rec = reinterpret_cast<float4 *>(a.impulses + (a.sort_order ? (uint64_t) a.sort_order[g] : g));
        const float4 vol = load_stream(rec + h), aux = load_stream(rec + h + 2);
        const uint32_t tag = dpp_u<QP_PAIR_HI>(__float_as_uint(aux.w));
        if (tag == 0u)
            continue;                             // ray had already escaped: slot keeps its zero fill
        uint32_t pair = 0;
        if (a.npairs > 1) {
            const float4 m4 = a.pair_mics[tag - 1u];
            mic = mk3(m4.x, m4.y, m4.z);
            pair = tag - 1u;
        }
        const float new_dist = dpp_f<QP_PAIR_HI>(aux.x), threshold = dpp_f<QP_PAIR_HI>(aux.y);
        const float4 * shade = reinterpret_cast<const float4 *>(a.scene.shade + dpp_u<QP_PAIR_HI>(__float_as_uint(aux.z)));
        const float4 sh = shade[0];
        const uint32_t skip_ref = __float_as_uint(shade[1].x);
        const uint32_t surface = __float_as_uint(sh.w);
        const v3 p = mk3(dpp_f<QP_PAIR_LO>(aux.x), dpp_f<QP_PAIR_LO>(aux.y), dpp_f<QP_PAIR_LO>(aux.z));
        const float diff = dpp_f<QP_PAIR_LO>(aux.w);
        const v3 b2p = mic - p;                   // kernel.cpp:282-286
        const float mag = length3(b2p);
        const v3 dir = normalize3(b2p);
        const uint32_t skip = fabsf(dot3(mk3(sh.x, sh.y, sh.z), dir)) > threshold ? skip_ref : RVB_BVH_EMPTY;
        const bool visible = !traverse_pair_any(a.scene, p, dir, mag, stack, skip);
        const float dist = visible ? new_dist + mag : 0.0f;          // kernel.cpp:471
        float4 o = make_float4(0, 0, 0, 0);
        if (visible) {
            float4 dc;                                               // diffuse coefficients of this lane's four bands
            if (SURF_LDS) dc = lds_load4(surf_lds, 4 * surface + 2 + h);
            else dc = reinterpret_cast<const float4 *>(a.scene.surfaces + surface)[2 + h];
            // kernel.cpp:480-485: newVol * attenuation * diffuse * DIFF, left to right
            o.x = ((vol.x * (air_attenuation(dist, air0) * 1.0f)) * dc.x) * diff;
            o.y = ((vol.y * (air_attenuation(dist, air1) * 1.0f)) * dc.y) * diff;
            o.z = ((vol.z * (air_attenuation(dist, air2) * 1.0f)) * dc.z) * diff;
            o.w = ((vol.w * (air_attenuation(dist, air3) * 1.0f)) * dc.w) * diff;
        }
        const float t = seconds_per_meter() * dist;                  // kernel.cpp:489
        store_stream(rec + h, o);
        store_stream(rec + h + 2, h == 0 ? make_float4(p.x, p.y, p.z, 0.0f) : make_float4(t, 0.0f, 0.0f, 0.0f));
        // inputs of findPredelay / MAX_SAMPLE (rayverb.h:49-74, rayverb.cpp:54-57): an impulse takes part iff any band is non-zero
        uint32_t nonzero = (o.x != 0.0f || o.y != 0.0f || o.z != 0.0f || o.w != 0.0f) ? 1u : 0u;
        nonzero |= dpp_u<QP_SWAP1>(nonzero);
        if (nonzero) {
            if (a.npairs > 1) {
                if (h == 0) {
                    const volatile uint32_t * seen = a.time_range + 2u * pair;
                    if (t != 0.0f && __float_as_uint(t) < seen[0]) atomicMin(a.time_range + 2u * pair, __float_as_uint(t));
                    if (__float_as_uint(t) > seen[1]) atomicMax(a.time_range + 2u * pair + 1u, __float_as_uint(t));
                }
            } else {
                if (t != 0.0f) tmin = fminf(tmin, t);
                tmax_seen = fmaxf(tmax_seen, t);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, off));
        tmax_seen = fmaxf(tmax_seen, __shfl_xor(tmax_seen, off));
    }
    if (threadIdx.x == 0 && a.npairs <= 1) {
        const volatile uint32_t * seen = a.time_range;
        if (tmin != __builtin_inff() && __float_as_uint(tmin) < seen[0]) atomicMin(a.time_range + 0, __float_as_uint(tmin));
        if (__float_as_uint(tmax_seen) > seen[1]) atomicMax(a.time_range + 1, __float_as_uint(tmax_seen));
    }
}

// shadow_pair_kernel with ONE lane per record (round 4, RVB_SHADOW_LANES=1): 64 records per wave pass, every lane walks its own any-hit
// query — four children and up to four triangles per step — with its own LDS stack column, nothing exchanged between lanes.  The records of
// a wave are neighbours in grouped order (same wall, same microphone), so unlike the path kernel's rays the lanes read mostly the SAME
// nodes: few distinct lines per load instruction.  Same operations on the same values as the pair kernel: same bytes
// (tests/test_gpu_parity.py::test_quad_shadow_kernel_gives_the_same_bytes runs it in a child process).  MEASURED at workload C2
// (profiles/r04_shadow_lanes_n1.txt): 2.12 ms against 1.26 (pairs) and 1.47 (quads) per 100 k rays x 128, the bench pipeline 5.40 against
// 4.42 ms per IR — like the one-lane path kernel it pays for its shorter instruction stream in 16-byte-per-lane loads (a leaf step alone is
// twelve of them per lane).  Kept for measurements only; the shipped form is two lanes per record.
template <bool SURF_LDS>
__global__ __launch_bounds__(WAVE, RVB_LANE_WAVES) void shadow_lane_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_lds[];   // [stack_entries + 1][64], surface table
    const uint32_t lane = threadIdx.x;
    const lds_float4_ptr surf_lds = stage_surfaces(a, stack_lds + (a.stack_entries + 1u) * LANE_RAYS);
    const uint64_t stride = (uint64_t) gridDim.x * LANE_RAYS, total = a.nrays * (uint64_t) a.nreflections;
    const char * node_base = reinterpret_cast<const char *>(a.scene.nodes);
    const char * tri_base = reinterpret_cast<const char *>(a.scene.tris);
    const float neg_cull = -a.scene.cull_abs;
    const lds_u32_ptr bottom = (lds_u32_ptr) stack_lds + lane;
    v3 mic = ld3(a.mic);
    float tmin = __builtin_inff(), tmax_seen = 0.0f;
    for (uint64_t g = (uint64_t) blockIdx.x * LANE_RAYS + lane; g < total; g += stride) {
        float4 * rec = reinterpret_cast<float4 *>(a.impulses + (a.sort_order ? (uint64_t) a.sort_order[g] : g));
        const float4 vol_lo = load_stream(rec + 0), vol_hi = load_stream(rec + 1), geo = load_stream(rec + 2), aux = load_stream(rec + 3);
        const uint32_t tag = __float_as_uint(aux.w);
        if (tag == 0u)
            continue;                             // ray had already escaped: slot keeps its zero fill
        uint32_t pair = 0;
        if (a.npairs > 1) {
            const float4 m4 = a.pair_mics[tag - 1u];
            mic = mk3(m4.x, m4.y, m4.z);
            pair = tag - 1u;
        }
        const float new_dist = aux.x, threshold = aux.y;
        const float4 * shade = reinterpret_cast<const float4 *>(a.scene.shade + __float_as_uint(aux.z));
        const float4 sh = shade[0];
        const uint32_t skip_ref = __float_as_uint(shade[1].x);
        const uint32_t surface = __float_as_uint(sh.w);
        const v3 p = mk3(geo.x, geo.y, geo.z);
        const float diff = geo.w;
        const v3 b2p = mic - p;                   // kernel.cpp:282-286
        const float mag = length3(b2p);
        const v3 dir = normalize3(b2p);
        const uint32_t skip = fabsf(dot3(mk3(sh.x, sh.y, sh.z), dir)) > threshold ? skip_ref : RVB_BVH_EMPTY;
        // any hit with EPSILON < distance <= mag? (traverse_pair_any with one lane: the lowest hit child is entered, the others pushed)
        bool blocked = false;
        {
            const float limit = fmaf(mag, 1.0f + a.scene.cull_rel, a.scene.cull_abs);
            const float ix = clamp_inv(dir.x), iy = clamp_inv(dir.y), iz = clamp_inv(dir.z);
            const float oix = p.x * ix, oiy = p.y * iy, oiz = p.z * iz;
            const uint32_t selx = slab_selector(ix), sely = slab_selector(iy), selz = slab_selector(iz);
            lds_u32_ptr sp = bottom;
            uint32_t ref = 0;
            for (;;) {
                while (!(ref & RVB_BVH_LEAF)) {
                    const uint4 * np = reinterpret_cast<const uint4 *>(node_base + ref);
                    const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                    float tn;
                    const bool ok0 = slab_select(n0, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn);
                    const bool ok1 = slab_select(n1, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn);
                    const bool ok2 = slab_select(n2, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn);
                    const bool ok3 = slab_select(n3, ix, iy, iz, oix, oiy, oiz, selx, sely, selz, limit, neg_cull, skip, tn);
                    // the lowest hit child is entered; the others go on the stack in child order (store, then advance if kept)
                    const bool first0 = ok0, first1 = ok1 && !ok0, first2 = ok2 && !(ok0 || ok1), first3 = ok3 && !(ok0 || ok1 || ok2);
                    *sp = n1.w; sp += (ok1 && !first1) ? LANE_RAYS : 0;
                    *sp = n2.w; sp += (ok2 && !first2) ? LANE_RAYS : 0;
                    *sp = n3.w; sp += (ok3 && !first3) ? LANE_RAYS : 0;
                    if (first0) ref = n0.w;
                    else if (first1) ref = n1.w;
                    else if (first2) ref = n2.w;
                    else if (first3) ref = n3.w;
                    else if (sp != bottom) { sp -= LANE_RAYS; ref = *sp; }
                    else ref = NONE;
                }
                if (ref == NONE)
                    break;
                const uint32_t first = ref & 0x0FFFFFFFu;
                const uint32_t count = ((ref >> 28) & 7u) + 1u;
                const float4 * tp0 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first));
                const float4 * tp1 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (1u < count ? 1u : 0u)));
                const float4 * tp2 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (2u < count ? 2u : 0u)));
                const float4 * tp3 = reinterpret_cast<const float4 *>(tri_base + tri_byte_offset(first + (3u < count ? 3u : 0u)));
                float4 ta = tp0[0], tb = tp0[1], tc = tp0[2], ua = tp1[0], ub = tp1[1], uc = tp1[2];
                float4 va = tp2[0], vb = tp2[1], vc = tp2[2], wa = tp3[0], wb = tp3[1], wc = tp3[2];
                asm volatile("" : "+v"(ta.x), "+v"(tb.x), "+v"(tc.x), "+v"(ua.x), "+v"(ub.x), "+v"(uc.x),
                                  "+v"(va.x), "+v"(vb.x), "+v"(vc.x), "+v"(wa.x), "+v"(wb.x), "+v"(wc.x));
                const float dist0 = mt_intersect(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), p, dir);
                const float dist1 = mt_intersect(mk3(ua.x, ua.y, ua.z), mk3(ua.w, ub.x, ub.y), mk3(ub.z, ub.w, uc.x), p, dir);
                const float dist2 = mt_intersect(mk3(va.x, va.y, va.z), mk3(va.w, vb.x, vb.y), mk3(vb.z, vb.w, vc.x), p, dir);
                const float dist3 = mt_intersect(mk3(wa.x, wa.y, wa.z), mk3(wa.w, wb.x, wb.y), mk3(wb.z, wb.w, wc.x), p, dir);
                if ((dist0 > RVB_EPSILON && dist0 <= mag) || (1u < count && dist1 > RVB_EPSILON && dist1 <= mag)
                    || (2u < count && dist2 > RVB_EPSILON && dist2 <= mag) || (3u < count && dist3 > RVB_EPSILON && dist3 <= mag)) {
                    blocked = true;
                    break;
                }
                if (sp != bottom) { sp -= LANE_RAYS; ref = *sp; } else break;
            }
        }
        const bool visible = !blocked;
        const float dist = visible ? new_dist + mag : 0.0f;          // kernel.cpp:471
        float4 o_lo = make_float4(0, 0, 0, 0), o_hi = o_lo;
        if (visible) {
            float4 d_lo, d_hi;                                       // diffuse coefficients
            if (SURF_LDS) { d_lo = lds_load4(surf_lds, 4 * surface + 2); d_hi = lds_load4(surf_lds, 4 * surface + 3); }
            else { const float4 * row = reinterpret_cast<const float4 *>(a.scene.surfaces + surface); d_lo = row[2]; d_hi = row[3]; }
            // kernel.cpp:480-485: newVol * attenuation * diffuse * DIFF, left to right
            o_lo.x = ((vol_lo.x * (air_attenuation(dist, a.air[0]) * 1.0f)) * d_lo.x) * diff;
            o_lo.y = ((vol_lo.y * (air_attenuation(dist, a.air[1]) * 1.0f)) * d_lo.y) * diff;
            o_lo.z = ((vol_lo.z * (air_attenuation(dist, a.air[2]) * 1.0f)) * d_lo.z) * diff;
            o_lo.w = ((vol_lo.w * (air_attenuation(dist, a.air[3]) * 1.0f)) * d_lo.w) * diff;
            o_hi.x = ((vol_hi.x * (air_attenuation(dist, a.air[4]) * 1.0f)) * d_hi.x) * diff;
            o_hi.y = ((vol_hi.y * (air_attenuation(dist, a.air[5]) * 1.0f)) * d_hi.y) * diff;
            o_hi.z = ((vol_hi.z * (air_attenuation(dist, a.air[6]) * 1.0f)) * d_hi.z) * diff;
            o_hi.w = ((vol_hi.w * (air_attenuation(dist, a.air[7]) * 1.0f)) * d_hi.w) * diff;
        }
        const float t = seconds_per_meter() * dist;                  // kernel.cpp:489
        store_stream(rec + 0, o_lo);
        store_stream(rec + 1, o_hi);
        store_stream(rec + 2, make_float4(p.x, p.y, p.z, 0.0f));
        store_stream(rec + 3, make_float4(t, 0.0f, 0.0f, 0.0f));
        const bool nonzero = o_lo.x != 0.0f || o_lo.y != 0.0f || o_lo.z != 0.0f || o_lo.w != 0.0f
                          || o_hi.x != 0.0f || o_hi.y != 0.0f || o_hi.z != 0.0f || o_hi.w != 0.0f;
        if (nonzero) {
            if (a.npairs > 1) {
                const volatile uint32_t * seen = a.time_range + 2u * pair;
                if (t != 0.0f && __float_as_uint(t) < seen[0]) atomicMin(a.time_range + 2u * pair, __float_as_uint(t));
                if (__float_as_uint(t) > seen[1]) atomicMax(a.time_range + 2u * pair + 1u, __float_as_uint(t));
            } else {
                if (t != 0.0f) tmin = fminf(tmin, t);
                tmax_seen = fmaxf(tmax_seen, t);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, off));
        tmax_seen = fmaxf(tmax_seen, __shfl_xor(tmax_seen, off));
    }
    if (threadIdx.x == 0 && a.npairs <= 1) {
        const volatile uint32_t * seen = a.time_range;
        if (tmin != __builtin_inff() && __float_as_uint(tmin) < seen[0]) atomicMin(a.time_range + 0, __float_as_uint(tmin));
        if (__float_as_uint(tmax_seen) > seen[1]) atomicMax(a.time_range + 1, __float_as_uint(tmax_seen));
    }
}

}  // namespace

// LDS of a quad kernel's single-wave workgroup: the traversal stack, then (optionally) the surface table, then — path kernel only — the key runs
static size_t quad_kernel_lds_bytes(const TraceArgs & a, bool key_runs = false)
{
    return a.stack_entries * QUADS_PER_BLOCK * sizeof(uint32_t) + (size_t) a.lds_surfaces * sizeof(rvb_surface) + RVB_LDS_NODES * 64u
           + (key_runs && a.sort_keys16 ? QUADS_PER_BLOCK * RVB_KEY_RUN * sizeof(uint16_t) : 0u);
}

// LDS of the two-lane path kernel's single-wave workgroup: stack, surface table, (experiment: top nodes), the lanes' cold words
static size_t rvb_pair_lds_bytes(const TraceArgs & a)
{
    return a.stack_entries * PAIRS_PER_BLOCK * sizeof(uint32_t) + (size_t) a.lds_surfaces * sizeof(rvb_surface) + RVB_LDS_NODES * 64u
           + (RVB_PAIR_COLD ? 5u * WAVE * sizeof(float) : 0u) + (a.sort_keys16 ? PAIRS_PER_BLOCK * RVB_KEY_RUN * sizeof(uint16_t) : 0u);
}

// LDS of the one-lane path kernel's single-wave workgroup: a stack column per lane (one slack row), surface table, a key run per lane
static size_t rvb_lane_lds_bytes(const TraceArgs & a)
{
    return (a.stack_entries + 1u) * LANE_RAYS * sizeof(uint32_t) + (size_t) a.lds_surfaces * sizeof(rvb_surface)
           + (a.sort_keys16 ? LANE_RAYS * RVB_KEY_RUN * sizeof(uint16_t) : 0u) + (RVB_LANE_COOP ? 4u * RVB_LANE_STAGE_STRIDE : 0u);
}

uint32_t rvb_lds_surfaces(uint32_t stack_entries, uint64_t nsurfaces)
{
    // 8 waves/SIMD = 32 single-wave workgroups per CU must still fit in the CU's 160 KiB of LDS: stack + key runs + surface table
    static const bool off = getenv("RVB_LDS_SURFACES") && getenv("RVB_LDS_SURFACES")[0] == '0';
    const size_t budget = (160u * 1024u) / 32u;
    const size_t stack = (size_t) stack_entries * QUADS_PER_BLOCK * sizeof(uint32_t) + QUADS_PER_BLOCK * RVB_KEY_RUN * sizeof(uint16_t);
    // ... and the two-lane kernels (twice the stack and key runs per workgroup, the cold words) want 5 waves/SIMD = 20 workgroups per CU
    const size_t pair_budget = (160u * 1024u) / 20u;
    const size_t pair_stack = (size_t) stack_entries * PAIRS_PER_BLOCK * sizeof(uint32_t) + PAIRS_PER_BLOCK * RVB_KEY_RUN * sizeof(uint16_t)
                              + (RVB_PAIR_COLD ? 5u * WAVE * sizeof(float) : 0u);
    if (off || nsurfaces == 0 || stack + nsurfaces * sizeof(rvb_surface) > budget || pair_stack + nsurfaces * sizeof(rvb_surface) > pair_budget)
        return 0;
    return (uint32_t) nsurfaces;
}

// Lanes per ray of the path kernel.  Per ray-bounce the pair kernel issues 17 % fewer VALU instructions than the quad kernel (the
// vote, stack and reduction work is per ray and every lane of the ray repeats it), but it has half the waves: it pays when the rays
// in flight fill the chip without the extra waves — one resident round of pair waves at 6 waves per SIMD is 6 x 1024 x 32 rays.
// Measured at workload C2 sizes (path kernel alone, ms per 100 k rays, quads / pairs): 100 k rays 3.60 / 3.79, 200 k 3.12 / 3.13,
// 400 k 2.77 / 2.49, 1 M 2.51 / 2.16; two 100 k traces in flight (the bench pipeline): 5.32 / 5.11 ms per IR.
uint32_t rvb_path_lanes_for(uint64_t nrays, uint32_t concurrent)
{
    static const int forced = getenv("RVB_PATH_LANES") ? atoi(getenv("RVB_PATH_LANES")) : 0;     // diagnostic override
    if (forced == 1 || forced == 2 || forced == 4) return (uint32_t) forced;
    // one lane per ray (path_lane_group_kernel) once the rays in flight give every SIMD RVB_LANE_MIN_WAVES waves of 64 rays: below
    // that its launch is bound by the latency of one wave's chain of (longer) steps, above it by how few instructions a ray costs
    static const uint64_t lane_min_waves = getenv("RVB_LANE_MIN_WAVES") ? strtoull(getenv("RVB_LANE_MIN_WAVES"), nullptr, 10) : 0;
    const uint64_t in_flight = nrays * (concurrent ? concurrent : 1u);
    if (lane_min_waves && in_flight >= lane_min_waves * 1024ull * LANE_RAYS) return 1u;
    return in_flight >= 6ull * 1024ull * PAIRS_PER_BLOCK ? 2u : 4u;
}

void rvb_launch_path(const TraceArgs & a, hipStream_t s)
{
    if (a.nrays == 0) return;
    if (a.path_lanes <= 2) {
        // one trace through the group kernel: the form that takes its arguments from the group block needs 80 registers and no scratch
        // (six waves per SIMD); a kernel of its own with TraceArgs by value came out at 86 once the key runs were added
        rvb_launch_path_group(&a, 1, s);
        return;
    }
    const unsigned blocks = (unsigned) ((a.nrays + QUADS_PER_BLOCK - 1) / QUADS_PER_BLOCK);
    const bool seven = a.nrays <= 7ull * 1024ull * QUADS_PER_BLOCK;      // fits in seven waves per SIMD: the 72-register build
    if (a.lds_surfaces) {
        if (seven) hipLaunchKernelGGL((path_kernel<true, 7>), dim3(blocks), dim3(WAVE), quad_kernel_lds_bytes(a, true), s, a);
        else hipLaunchKernelGGL((path_kernel<true, 8>), dim3(blocks), dim3(WAVE), quad_kernel_lds_bytes(a, true), s, a);
    } else {
        if (seven) hipLaunchKernelGGL((path_kernel<false, 7>), dim3(blocks), dim3(WAVE), quad_kernel_lds_bytes(a, true), s, a);
        else hipLaunchKernelGGL((path_kernel<false, 8>), dim3(blocks), dim3(WAVE), quad_kernel_lds_bytes(a, true), s, a);
    }
}

// (the caller checked: every trace has the same lane count, stack depth, number of surfaces staged in LDS and key form)
void rvb_launch_path_group(const TraceArgs * traces, uint32_t count, hipStream_t s)
{
    TraceGroup g;
    g.count = count;
    const TraceArgs & a = traces[0];
    const uint32_t rays_per_block = a.path_lanes == 1 ? LANE_RAYS : PAIRS_PER_BLOCK;
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < count; ++k) {
        g.first_block[k] = blocks;
        g.trace[k] = traces[k];
        blocks += (uint32_t) ((traces[k].nrays + rays_per_block - 1) / rays_per_block);
    }
    for (uint32_t k = count; k <= RVB_MAX_GROUP; ++k) g.first_block[k] = blocks;
    for (uint32_t k = count; k < RVB_MAX_GROUP; ++k) g.trace[k] = traces[0];
    size_t lds = 0;                          // (the largest of the traces', should a caller ever group traces whose layouts differ in size)
    for (uint32_t k = 0; k < count; ++k) lds = std::max(lds, a.path_lanes == 1 ? rvb_lane_lds_bytes(traces[k]) : rvb_pair_lds_bytes(traces[k]));
    // measurement: an LDS request per workgroup that caps the path waves per CU (160 KiB / bytes), leaving wave slots and registers to other kernels
    static const size_t lds_floor = getenv("RVB_PATH_LDS_BYTES") ? strtoull(getenv("RVB_PATH_LDS_BYTES"), nullptr, 10) : 0;
    lds = std::max(lds, lds_floor);
    if (a.path_lanes == 1) {
        if (a.lds_surfaces) hipLaunchKernelGGL(path_lane_group_kernel<true>, dim3(blocks), dim3(WAVE), lds, s, g);
        else hipLaunchKernelGGL(path_lane_group_kernel<false>, dim3(blocks), dim3(WAVE), lds, s, g);
        return;
    }
    if (a.lds_surfaces) hipLaunchKernelGGL(path_pair_group_kernel<true>, dim3(blocks), dim3(WAVE), lds, s, g);
    else hipLaunchKernelGGL(path_pair_group_kernel<false>, dim3(blocks), dim3(WAVE), lds, s, g);
}

void rvb_launch_images(const TraceArgs & a, hipStream_t s)
{
    const unsigned blocks = (unsigned) ((a.nrays + WAVE - 1) / WAVE);     // one lane per ray
    hipLaunchKernelGGL(image_plan_kernel, dim3(blocks ? blocks : 1), dim3(WAVE), 0, s, a);
    // a few hundred list entries at workload C2: 256 single-wave workgroups of 16 quads walk the list whatever its length
    hipLaunchKernelGGL(image_check_kernel, dim3(256), dim3(WAVE), a.stack_entries * QUADS_PER_BLOCK * sizeof(uint32_t), s, a);
}

// Two lanes per record by default (shadow_pair_kernel): 12.8 M records fill the chip whatever the lane count, and a record costs
// 12 % less (C2: 1.45 -> 1.28 ms).  RVB_SHADOW_LANES=4 keeps the quad kernel (measurements).
uint32_t rvb_shadow_lanes()
{
    static const int lanes = getenv("RVB_SHADOW_LANES") ? atoi(getenv("RVB_SHADOW_LANES")) : 2;
    return lanes == 4 ? 4u : (lanes == 1 ? 1u : 2u);
}

void rvb_launch_shadow(const TraceArgs & a, hipStream_t s)
{
    const uint64_t total = a.nrays * (uint64_t) a.nreflections;
    if (total == 0) return;
    uint64_t blocks = (total + QUADS_PER_BLOCK - 1) / QUADS_PER_BLOCK;
    static const uint64_t per_cu = getenv("RVB_SHADOW_WG_PER_CU") ? strtoull(getenv("RVB_SHADOW_WG_PER_CU"), nullptr, 10) : 256;
    if (blocks > 256u * per_cu) blocks = 256u * per_cu;     // single-wave workgroups per CU; records beyond are grid-strided
    if (rvb_shadow_lanes() == 1) {
        blocks = (total + LANE_RAYS - 1) / LANE_RAYS;
        static const uint64_t lane_per_cu = getenv("RVB_SHADOW_WG_PER_CU") ? strtoull(getenv("RVB_SHADOW_WG_PER_CU"), nullptr, 10) : 128;
        if (blocks > 256u * lane_per_cu) blocks = 256u * lane_per_cu;
        const size_t lds = (a.stack_entries + 1u) * LANE_RAYS * sizeof(uint32_t) + (size_t) a.lds_surfaces * sizeof(rvb_surface);
        if (a.lds_surfaces) hipLaunchKernelGGL(shadow_lane_kernel<true>, dim3((unsigned) blocks), dim3(WAVE), lds, s, a);
        else hipLaunchKernelGGL(shadow_lane_kernel<false>, dim3((unsigned) blocks), dim3(WAVE), lds, s, a);
        return;
    }
    if (rvb_shadow_lanes() == 2) {
        blocks = (total + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK;
        static const uint64_t pair_per_cu = getenv("RVB_SHADOW_WG_PER_CU") ? strtoull(getenv("RVB_SHADOW_WG_PER_CU"), nullptr, 10) : 256;
        if (blocks > 256u * pair_per_cu) blocks = 256u * pair_per_cu;
        const size_t lds = a.stack_entries * PAIRS_PER_BLOCK * sizeof(uint32_t) + (size_t) a.lds_surfaces * sizeof(rvb_surface);
        if (a.lds_surfaces) hipLaunchKernelGGL(shadow_pair_kernel<true>, dim3((unsigned) blocks), dim3(WAVE), lds, s, a);
        else hipLaunchKernelGGL(shadow_pair_kernel<false>, dim3((unsigned) blocks), dim3(WAVE), lds, s, a);
        return;
    }
    if (a.lds_surfaces) hipLaunchKernelGGL(shadow_kernel<true>, dim3((unsigned) blocks), dim3(WAVE), quad_kernel_lds_bytes(a), s, a);
    else hipLaunchKernelGGL(shadow_kernel<false>, dim3((unsigned) blocks), dim3(WAVE), quad_kernel_lds_bytes(a), s, a);
}
