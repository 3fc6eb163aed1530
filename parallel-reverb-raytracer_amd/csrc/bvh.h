// bvh.h — acceleration structure of the trace kernels (new design: the reference has none and
// scans all triangles per query, reference rayverb/kernel.cpp:167-192).
//
// The BVH only decides WHICH triangles get tested; every candidate is tested with the
// reference's own Möller–Trumbore arithmetic (rvb_math.h mt_intersect) and the winner is chosen
// by the reference's rule (smallest distance > EPSILON, ties to the lowest triangle index,
// kernel.cpp:180-188), so a query returns what the brute-force scan returns.  Boxes are padded
// and the cull test carries slack (see kPad / cull_slack in trace_kernels.hip) because the
// float result of the triangle test can land slightly outside the true triangle / true distance.
//
// HBM layout (all arrays read-only during a trace, resident in L2 / Infinity Cache):
//   nodes   : BvhNode[],    64 B, 4-wide, breadth-first (top levels contiguous); one 16-byte
//             record per child — box as six binary16 values rounded OUTWARD (lo down, hi up), so the
//             four lanes that cooperate on one ray (trace_kernels.hip) read one contiguous 64-byte
//             half line per node visit with a single 16-byte load each.  Boxes only prune, so their
//             precision does not touch results; halving node bytes halves the L1 (TCP) traffic.
//   tris    : BvhTri[],     48 B, in leaf order: v0, e0, e1 (edges precomputed), original index
//   shade   : TriShade[],   32 B, by ORIGINAL triangle index: unit normal + surface index + own-plane skip
//   verts9  : TriCorners[], 48 B, by ORIGINAL triangle index: the three vertices (image-source)
#pragma once

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/rvb_capi.h"

#define RVB_BVH_EMPTY 0xFFFFFFFFu
#define RVB_BVH_LEAF 0x80000000u
#define RVB_BVH_NODE_SHIFT 6         // node references are BYTE offsets (index * 64): one 32-bit add forms the load address
#ifndef RVB_BVH_MAX_LEAF
#define RVB_BVH_MAX_LEAF 4
#endif
#define RVB_BVH_STACK 64            // per-lane traversal stack entries (LDS)

struct BvhChild {                   // 16 B = one 16-byte load
    uint16_t lox, hix, loy, hiy, loz, hiz;   // binary16 bit patterns, the two planes of an axis in one 32-bit word (a conditional
                                    // swap of its halves puts the plane the ray meets first in the low half: slab_select in
                                    // trace_kernels.hip).  An EMPTY slot holds lo = +inf, hi = -inf.
    uint32_t ref;                   // EMPTY | LEAF|(count-1)<<28|first | node index << RVB_BVH_NODE_SHIFT
};
struct BvhNode { BvhChild c[4]; };  // 64 B

struct BvhTri {                     // 48 B
    float v0[3];
    float e0[3];
    float e1[3];
    uint32_t index;                 // original triangle index (tie-break + results)
    uint32_t surface;
    uint32_t pad;
};

struct TriCorners { float v[9]; float pad[3]; };       // 48 B

// Shading record by ORIGINAL triangle index, 32 B = two 16-byte loads from one half line: unit normal + surface, and the
// own-plane skip of the triangle.
//
// Own-plane skip.  A ray that STARTS on triangle T (a reflected ray, or the shadow ray of the same point) cannot hit — with a
// distance above EPSILON — any triangle that lies in T's plane, provided it leaves the plane steeply enough: such a triangle's
// Möller–Trumbore distance is (offset of the origin from the plane) / |cos|, i.e. rounding noise divided by |cos|.  `skip_ref`
// is the child reference of the LARGEST subtree around T whose triangles are all coplanar with T (RVB_BVH_EMPTY if there is
// none; the builder makes every plane patch a subtree of its own so that a wall is one such child); the traversal rejects that
// child without visiting it when
//      |dot(unit normal of T, direction)|  >  skip_a + skip_b * (length of the ray segment that ended on T)
// skip_a, skip_b come from the error analysis in bvh_build.hip (assign_skips).  Without the skip every query first descends
// into the wall it starts on (the padded boxes around its origin cannot be culled).
struct TriShade { float n[3]; uint32_t surface; uint32_t skip_ref; float skip_a; float skip_b; uint32_t group; };     // 32 B; on the device `group` holds the triangle's leaf position (capi.hip)

struct BuiltScene {
    std::vector<BvhNode> nodes;
    std::vector<BvhTri> tris;
    std::vector<TriShade> shade;
    std::vector<TriCorners> corners;
    std::vector<uint32_t> leafpos;  // by ORIGINAL triangle index: position in `tris` (spatially coherent), 0xFFFFFFFF if dropped
    uint32_t depth = 0;             // levels of 4-wide nodes
    uint32_t stack_need = 0;        // worst-case traversal stack entries (<= RVB_BVH_STACK)
    float pad = 0.0f;               // box padding actually used (metres)
    float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
};

// Returns an empty string on success, else the error text.
std::string rvb_build_scene(const rvb_triangle * triangles, uint64_t ntriangles,
                            const rvb_float3 * vertices, uint64_t nvertices,
                            uint64_t nsurfaces, BuiltScene & out);
