// bvh_build.hip — host-side scene preparation for rvb_set_scene: validation, per-triangle
// precomputation (edges / normals with the kernels' own arithmetic) and a binned-SAH binary BVH
// collapsed to 4-wide nodes in breadth-first order.  Host code only (no kernels here).
#include "bvh.h"
#include "rvb_math.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>

namespace {

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
    }
    void grow(const float p[3])
    {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    void grow(const Box & b)
    {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    float area() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f && dy >= 0.0f && dz >= 0.0f))
            return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct Prim { Box box; float c[3]; uint32_t tri; };

struct BinNode {
    Box box;
    int32_t left = -1, right = -1;     // children (internal) ...
    uint32_t first = 0, count = 0;     // ... or primitive range (leaf)
    bool leaf() const { return left < 0; }
};

// float -> binary16 bit pattern, rounded toward -inf (down = true) or +inf: conservative box planes
uint16_t half_bits(_Float16 h)
{
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
_Float16 half_from_bits(uint16_t b)
{
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return h;
}
uint16_t to_half_outward(float x, bool down)
{
    _Float16 h = (_Float16) x;                       // round to nearest even (may overflow to +-inf)
    uint16_t b = half_bits(h);
    const float back = (float) h;
    if (down ? back > x : back < x) {                // step one binary16 value in the outward direction
        if (down) {
            if ((b & 0x7FFFu) == 0) b = 0x8001u;      // +-0 -> smallest negative
            else if (b & 0x8000u) b = (uint16_t) (b + 1);   // negative: larger magnitude
            else b = (uint16_t) (b - 1);              // positive: smaller magnitude
        } else {
            if ((b & 0x7FFFu) == 0) b = 0x0001u;
            else if (b & 0x8000u) b = (uint16_t) (b - 1);
            else b = (uint16_t) (b + 1);
        }
    }
    (void) half_from_bits;
    return b;
}

#ifndef RVB_SAH_BINS
#define RVB_SAH_BINS 16
#endif
const int kBins = RVB_SAH_BINS;
const int kMaxBinaryDepth = 48;

struct Builder {
    std::vector<Prim> prims;
    std::vector<BinNode> nodes;
    bool always_median = false;         // balanced fallback when the SAH tree needs too deep a stack

    int build(uint32_t first, uint32_t count, int depth)
    {
        BinNode node;
        node.box.reset();
        Box cbox;
        cbox.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            node.box.grow(prims[i].box);
            cbox.grow(prims[i].c);
        }
        int id = (int) nodes.size();
        nodes.push_back(node);

        auto make_leaf = [&]() {
            nodes[id].first = first;
            nodes[id].count = count;
            return id;
        };
        if (count <= 1)
            return make_leaf();

        int axis = 0;
        float ext[3];
        for (int a = 0; a < 3; ++a)
            ext[a] = cbox.hi[a] - cbox.lo[a];
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;

        // depth budget: fall back to a median split when SAH could not finish within it
        int needed = 0;
        for (uint32_t c = count; c > RVB_BVH_MAX_LEAF; c = (c + 1) / 2)
            ++needed;
        bool force_median = always_median || depth + needed + 2 >= kMaxBinaryDepth;

        uint32_t mid = first + count / 2;
        bool have_split = false;
        if (!force_median && ext[axis] > 0.0f) {
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; ++a) {
                if (!(ext[a] > 0.0f))
                    continue;
                Box bb[kBins];
                uint32_t bc[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
                float scale = (float) kBins / ext[a];
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = std::min(kBins - 1, std::max(0, (int) ((prims[i].c[a] - cbox.lo[a]) * scale)));
                    bb[b].grow(prims[i].box);
                    ++bc[b];
                }
                float right_area[kBins];
                uint32_t right_count[kBins];
                Box acc;
                acc.reset();
                uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bb[b]);
                    n += bc[b];
                    right_area[b] = acc.area();
                    right_count[b] = n;
                }
                acc.reset();
                n = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bb[b]);
                    n += bc[b];
                    if (n == 0 || right_count[b + 1] == 0)
                        continue;
                    float cost = acc.area() * (float) n + right_area[b + 1] * (float) right_count[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            float leaf_cost = node.box.area() * (float) count;
            // Four lanes test the (up to four) triangles of a leaf in one step, so a full leaf costs what a
            // single-triangle leaf costs: the default fills leaves (measured on C2: path 5.57 -> 5.18 ms), RVB_LEAF_POLICY=sah keeps the SAH decision.
            static const bool sah_leaves = getenv("RVB_LEAF_POLICY") && std::string(getenv("RVB_LEAF_POLICY")) == "sah";
            bool want_leaf = count <= RVB_BVH_MAX_LEAF && (!sah_leaves || !(best_cost + node.box.area() < leaf_cost));
            if (want_leaf)
                return make_leaf();
            if (best_axis >= 0) {
                float scale = (float) kBins / ext[best_axis];
                float lo = cbox.lo[best_axis];
                auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim & p) {
                    int b = std::min(kBins - 1, std::max(0, (int) ((p.c[best_axis] - lo) * scale)));
                    return b <= best_bin;
                });
                mid = (uint32_t) (it - prims.begin());
                have_split = mid > first && mid < first + count;
            }
        }
        if (!have_split) {
            if (count <= RVB_BVH_MAX_LEAF)
                return make_leaf();
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [axis](const Prim & a, const Prim & b) {
                                 if (a.c[axis] != b.c[axis]) return a.c[axis] < b.c[axis];
                                 return a.tri < b.tri;
                             });
        }
        int l = build(first, mid - first, depth + 1);
        int r = build(mid, first + count - mid, depth + 1);
        nodes[id].left = l;
        nodes[id].right = r;
        return id;
    }
};

}  // namespace

std::string rvb_build_scene(const rvb_triangle * triangles, uint64_t ntriangles,
                            const rvb_float3 * vertices, uint64_t nvertices,
                            uint64_t nsurfaces, BuiltScene & out)
{
    if (ntriangles >= (1ull << 24))
        return "too many triangles (limit 2^24: triangle byte offsets are formed with a 24-bit multiply, node references are 31-bit byte offsets)";
    out = BuiltScene();
    out.shade.resize(ntriangles);
    out.corners.resize(ntriangles);

    float maxabs = 0.0f;
    Box scene;
    scene.reset();
    for (uint64_t i = 0; i < nvertices; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = vertices[i].s[a];
            if (std::isfinite(v)) {
                maxabs = std::max(maxabs, std::fabs(v));
                scene.lo[a] = std::min(scene.lo[a], v);
                scene.hi[a] = std::max(scene.hi[a], v);
            }
        }
    for (int a = 0; a < 3; ++a) { out.bounds_lo[a] = scene.lo[a]; out.bounds_hi[a] = scene.hi[a]; }
    // keeps every Möller–Trumbore determinant below 2^64, the range reciprocal_cr (rvb_math.h) is verified on
    if (maxabs > 0x1p30f)
        return "vertex coordinates beyond 2^30 are not supported";
    // Box padding: the float Möller–Trumbore test accepts hits a little outside the exact
    // triangle; 1 mm + 2e-5 of the coordinate range covers that with a wide margin.
    const float pad = std::max(1e-3f, 2e-5f * maxabs);
    out.pad = pad;

    Builder b;
    b.prims.reserve(ntriangles);
    for (uint64_t i = 0; i < ntriangles; ++i) {
        const rvb_triangle & t = triangles[i];
        if (t.surface >= nsurfaces || t.v0 >= nvertices || t.v1 >= nvertices || t.v2 >= nvertices)
            return "triangle " + std::to_string(i) + " refers to a surface or vertex out of range";
        v3 p0 = mk3(vertices[t.v0].s[0], vertices[t.v0].s[1], vertices[t.v0].s[2]);
        v3 p1 = mk3(vertices[t.v1].s[0], vertices[t.v1].s[1], vertices[t.v1].s[2]);
        v3 p2 = mk3(vertices[t.v2].s[0], vertices[t.v2].s[1], vertices[t.v2].s[2]);
        TriVerts tv = {p0, p1, p2};
        v3 n = verts_normal(tv);                     // reference kernel.cpp:109-127
        out.shade[i].n[0] = n.x; out.shade[i].n[1] = n.y; out.shade[i].n[2] = n.z;
        out.shade[i].surface = (uint32_t) t.surface;
        float c9[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
        std::memcpy(out.corners[i].v, c9, sizeof(c9));
        std::memset(out.corners[i].pad, 0, sizeof(out.corners[i].pad));

        bool finite = true;
        for (float f : c9) finite = finite && std::isfinite(f);
        if (!finite)
            continue;   // NaN/inf vertices: every comparison of the triangle test fails -> never hit
        // |det| <= |e0 x e1| for a unit direction; below EPSILON the triangle can never be hit
        // (quirk Q7).  Evaluated in double with a 1 % margin so nothing hittable is dropped.
        double e0[3] = {(double) p1.x - p0.x, (double) p1.y - p0.y, (double) p1.z - p0.z};
        double e1[3] = {(double) p2.x - p0.x, (double) p2.y - p0.y, (double) p2.z - p0.z};
        double cx = e0[1] * e1[2] - e0[2] * e1[1], cy = e0[2] * e1[0] - e0[0] * e1[2], cz = e0[0] * e1[1] - e0[1] * e1[0];
        if (std::sqrt(cx * cx + cy * cy + cz * cz) * 1.01 < (double) RVB_EPSILON)
            continue;

        Prim p;
        p.tri = (uint32_t) i;
        p.box.reset();
        p.box.grow(&c9[0]); p.box.grow(&c9[3]); p.box.grow(&c9[6]);
        for (int a = 0; a < 3; ++a) {
            p.c[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
            p.box.lo[a] -= pad;
            p.box.hi[a] += pad;
        }
        b.prims.push_back(p);
    }

    const uint32_t nprims = (uint32_t) b.prims.size();
    if (nprims == 0) {
        BvhNode root;
        std::memset(&root, 0, sizeof(root));
        for (int c = 0; c < 4; ++c) {
            root.c[c].ref = RVB_BVH_EMPTY;
            root.c[c].lox = root.c[c].loy = root.c[c].loz = root.c[c].hix = root.c[c].hiy = root.c[c].hiz = 0x7E00u;   // NaN
        }
        out.nodes.push_back(root);
        out.depth = 1;
        return "";
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
    b.always_median = attempt == 1;
    b.nodes.clear();
    b.nodes.reserve(2 * (size_t) nprims);
    out.nodes.clear();
    int root = b.build(0, nprims, 0);

    // leaf-order triangle records
    out.tris.resize(nprims);
    for (uint32_t i = 0; i < nprims; ++i) {
        uint32_t ti = b.prims[i].tri;
        const float * c = out.corners[ti].v;
        v3 p0 = mk3(c[0], c[1], c[2]), p1 = mk3(c[3], c[4], c[5]), p2 = mk3(c[6], c[7], c[8]);
        v3 e0 = p1 - p0, e1 = p2 - p0;               // reference kernel.cpp:65-66
        BvhTri & t = out.tris[i];
        t.v0[0] = p0.x; t.v0[1] = p0.y; t.v0[2] = p0.z;
        t.e0[0] = e0.x; t.e0[1] = e0.y; t.e0[2] = e0.z;
        t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
        t.index = ti;
        t.surface = out.shade[ti].surface;
        t.pad = 0;
    }
    // triangles of one leaf in ascending original index (keeps the tie rule cheap to reason about)
    for (const BinNode & n : b.nodes)
        if (n.leaf())
            std::sort(out.tris.begin() + n.first, out.tris.begin() + n.first + n.count,
                      [](const BvhTri & x, const BvhTri & y) { return x.index < y.index; });

    // collapse to 4-wide, breadth-first
    struct Item { int bin; uint32_t slot; uint32_t depth; };
    std::queue<Item> q;
    out.nodes.emplace_back();
    q.push({root, 0, 1});
    uint32_t max_depth = 1;
    while (!q.empty()) {
        Item it = q.front();
        q.pop();
        max_depth = std::max(max_depth, it.depth);
        int kids[4];
        int nk = 0;
        if (b.nodes[it.bin].leaf()) {
            kids[nk++] = it.bin;
        } else {
            kids[nk++] = b.nodes[it.bin].left;
            kids[nk++] = b.nodes[it.bin].right;
            while (nk < 4) {
                int pick = -1;
                float pick_area = -1.0f;
                for (int k = 0; k < nk; ++k)
                    if (!b.nodes[kids[k]].leaf() && b.nodes[kids[k]].box.area() > pick_area) {
                        pick = k;
                        pick_area = b.nodes[kids[k]].box.area();
                    }
                if (pick < 0)
                    break;
                int expand = kids[pick];
                kids[pick] = b.nodes[expand].left;
                kids[nk++] = b.nodes[expand].right;
            }
        }
        BvhNode node;
        std::memset(&node, 0, sizeof(node));
        for (int k = 0; k < 4; ++k) {
            BvhChild & slot = node.c[k];
            if (k >= nk) {
                slot.ref = RVB_BVH_EMPTY;
                // empty slot: rejected by its ref in the slab test; the box is never used
                slot.lox = slot.loy = slot.loz = slot.hix = slot.hiy = slot.hiz = 0x7E00u;
                continue;
            }
            const BinNode & c = b.nodes[kids[k]];
            slot.lox = to_half_outward(c.box.lo[0], true); slot.loy = to_half_outward(c.box.lo[1], true);
            slot.loz = to_half_outward(c.box.lo[2], true);
            slot.hix = to_half_outward(c.box.hi[0], false); slot.hiy = to_half_outward(c.box.hi[1], false);
            slot.hiz = to_half_outward(c.box.hi[2], false);
            if (c.leaf()) {
                if (c.count > RVB_BVH_MAX_LEAF)
                    return "internal error: oversized BVH leaf";
                slot.ref = RVB_BVH_LEAF | ((c.count - 1) << 28) | c.first;
            } else {
                uint32_t slot_index = (uint32_t) out.nodes.size();
                out.nodes.emplace_back();
                node.c[k].ref = slot_index << RVB_BVH_NODE_SHIFT;     // byte offset of the child node
                q.push({kids[k], slot_index, it.depth + 1});
            }
        }
        out.nodes[it.slot] = node;
    }
    out.depth = max_depth;
    out.leafpos.assign(ntriangles, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < nprims; ++i)
        out.leafpos[out.tris[i].index] = i;
    // Worst-case traversal stack: descending into one child leaves the other children of the
    // node on the stack.  Nodes are breadth-first, so children have larger indices than parents.
    std::vector<uint32_t> need(out.nodes.size(), 0);
    for (size_t i = out.nodes.size(); i-- > 0;) {
        uint32_t nchild = 0, deepest = 0;
        for (int k = 0; k < 4; ++k) {
            uint32_t c = out.nodes[i].c[k].ref;
            if (c == RVB_BVH_EMPTY)
                continue;
            ++nchild;
            if (!(c & RVB_BVH_LEAF))
                deepest = std::max(deepest, need[c >> RVB_BVH_NODE_SHIFT]);
        }
        need[i] = (nchild ? nchild - 1 : 0) + deepest;
    }
    out.stack_need = need[0] + 1;
    if (out.stack_need <= RVB_BVH_STACK)
        return "";
    }
    return "BVH needs a deeper traversal stack than " + std::to_string(RVB_BVH_STACK) + " entries";
}
