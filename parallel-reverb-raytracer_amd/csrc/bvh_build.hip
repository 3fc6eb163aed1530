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
#include <tuple>

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

// unit: the plane patch the triangle belongs to when that patch is built as a subtree of its own (kLoose otherwise);
// key: the centroid the builder splits on — the unit's while several units share a node, the triangle's own inside its unit
const uint32_t kLoose = 0xFFFFFFFFu;
struct Prim { Box box; float c[3]; uint32_t tri; uint32_t unit = kLoose; float key[3] = {0, 0, 0}; float ukey[3] = {0, 0, 0}; };

struct BinNode {
    Box box;
    int32_t left = -1, right = -1;     // children (internal) ...
    uint32_t first = 0, count = 0;     // ... or primitive range (leaf)
    bool unit_root = false;            // root of a plane patch's own subtree: stays ONE child slot of its 4-wide parent
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

    // Plane patches (units) are kept whole: a node either lies inside one unit or holds whole units (and loose triangles)
    // only.  While several units share a node every triangle is binned at its UNIT's centroid, so a split never cuts a unit;
    // the node that holds exactly one unit is that unit's root (BinNode::unit_root) and splits on the triangles' own
    // centroids from there on.  This is what makes a wall ONE child slot that a ray starting on it can skip (TriShade::skip_ref).
    int build(uint32_t first, uint32_t count, int depth, bool inside_unit = false)
    {
        BinNode node;
        node.box.reset();
        bool unit_root = false;
        if (!inside_unit && prims[first].unit != kLoose) {
            unit_root = true;
            for (uint32_t i = first + 1; i < first + count && unit_root; ++i)
                unit_root = prims[i].unit == prims[first].unit;
            if (unit_root) {
                inside_unit = true;
                for (uint32_t i = first; i < first + count; ++i)
                    for (int a = 0; a < 3; ++a) prims[i].key[a] = prims[i].c[a];
            }
        }
        node.unit_root = unit_root;
        Box cbox;
        cbox.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            node.box.grow(prims[i].box);
            cbox.grow(prims[i].key);
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
                    int b = std::min(kBins - 1, std::max(0, (int) ((prims[i].key[a] - cbox.lo[a]) * scale)));
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
                    int b = std::min(kBins - 1, std::max(0, (int) ((p.key[best_axis] - lo) * scale)));
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
            auto by_key = [axis](const Prim & a, const Prim & b) {
                if (a.key[axis] != b.key[axis]) return a.key[axis] < b.key[axis];
                if (a.unit != b.unit) return a.unit < b.unit;
                return a.tri < b.tri;
            };
            if (inside_unit) {
                std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count, by_key);
            } else {
                // several units: the cut goes to the unit boundary nearest to the middle
                std::sort(prims.begin() + first, prims.begin() + first + count, by_key);
                auto boundary = [&](uint32_t i) { return prims[i - 1].unit == kLoose || prims[i].unit == kLoose || prims[i - 1].unit != prims[i].unit; };
                uint32_t best = 0;
                for (uint32_t i = first + 1; i < first + count; ++i)
                    if (boundary(i) && (best == 0 || (i > mid ? i - mid : mid - i) < (best > mid ? best - mid : mid - best))) best = i;
                mid = best;        // exists: the node holds more than one unit / loose triangle
            }
        }
        int l = build(first, mid - first, depth + 1, inside_unit);
        int r = build(mid, first + count - mid, depth + 1, inside_unit);
        nodes[id].left = l;
        nodes[id].right = r;
        return id;
    }
};


// ---- own-plane skip (the skip fields of TriShade, bvh.h) ------------------------------------------------------------
// Claim.  Let the ray (o, d) start on triangle T: o = fl(o_prev + d_prev * t) with t the Möller–Trumbore distance of T for
// (o_prev, d_prev) (path_kernel), or the same stored point (shadow_kernel).  Let T' be a triangle whose three effective
// vertices (v0, v0 + e0, v0 + e1 with the float edges the kernels use) and those of T lie within delta of one plane P, with
// unit normals within theta of P's.  Then the distance mt_intersect returns for (T', o, d), if it returns one at all, obeys
//      |t'| <= (h + 16 eps |tvec'| |e0'| |e1'| / |N'|) (1 + 3 eps) / (c' - 16 eps kappa')           [eps = 2^-24]
// where N' = e0' x e1', kappa' = |e0'| |e1'| / |N'|, c' = |d . n'| and h = distance of o from the plane of T':
//   * numerator e1 . (tvec x e0) = tvec . N' = h |N'| exactly; evaluated in binary32, one rounding per operator, its error is
//     at most 8 eps |tvec| |e0| |e1| (cross: 2 eps per component pair, dot: 3 eps, the rounding of tvec = o - v0: 1 eps);
//     16 eps is used.  The determinant e0 . (d x e1) = -d . N' likewise, relative to |d| |e0| |e1|.
//   * a returned distance has passed 0 <= u, v <= 1 (computed), so tvec' = u e0' + v e1' - t' d and |tvec'| <= 1.1 (L' + |t'|)
//     with L' = |e0'| + |e1'| (the computed u, v are within 8 eps kappa' |tvec'| / c' of the exact ones; kappa' <= 100 and
//     c' >= 0.01 are enforced below).
//   * o lies inside T up to rounding, so by convexity its offset from the plane of T' is at most the offset from the plane of
//     T plus 6 delta; its offset from the plane of T is the error of the hit point: |t_c - t*| c + rounding of the three
//     components = 16 eps kappa (t + 2 L) + 4 eps t + sqrt(3) eps max|coordinate|.
// So |t'| <= EPSILON — no hit, reference kernel.cpp:180 — whenever
//      c > C + (A + B t) / EPSILON,   A = 6 delta + G + 32 eps kappa L + 1.74 eps maxabs,  B = 16 eps kappa + 4 eps,
//      G = max over the group of 34 eps (L' + 0.02) kappa',  C = 2 theta + max 16 eps kappa' + 8 eps,
// with c = |dot(stored unit normal of T, d)| as the kernels compute it.  Constants carry a factor 2 over the derivation and the
// final inequality another 1e-4 relative; TriShade::skip_a = C + A (1 + 1e-4) / EPSILON (at least 0.01), skip_b = B (1 + 1e-4) / EPSILON.
// tools/travsim.cpp (TRAVSIM_VERIFY) checks the rule against brute force on the host, the GPU parity tests on the device.
struct PlaneGroup {
    double n[3] = {0, 0, 1}, p[3] = {0, 0, 0};      // reference plane (that of the group's largest triangle): unit normal, point
    double area = -1.0, delta = 0.0, theta = 0.0, g = 0.0, d = 0.0;
    uint32_t size = 0;
    bool ok = true;
};

struct TriGeo { double n[3], v[3][3], area, kappa, L; bool usable; };

struct Planes {
    std::vector<TriGeo> geo;          // by original triangle index (kept triangles only)
    std::vector<uint32_t> gid;        // by original triangle index: plane group, 0xFFFFFFFF = none
    std::vector<PlaneGroup> groups;
};

// Plane groups = connected patches of coplanar triangles: two kept triangles that share a vertex position are united when
// their planes agree (normals within 1e-4 up to sign, each one's vertices within 5 um of the other's plane).  The group-level
// tolerances (delta, theta against the plane of the group's largest triangle) are checked on the result: a finely curved
// surface can chain, its group is then not `ok` and takes no part in anything.
void find_planes(const BuiltScene & out, const std::vector<Prim> & prims, Planes & pl)
{
    const double eps = 0x1p-24;
    const size_t ntri = out.shade.size();
    pl.geo.assign(ntri, TriGeo());
    pl.gid.assign(ntri, 0xFFFFFFFFu);
    pl.groups.clear();
    std::vector<uint32_t> parent(ntri);
    for (size_t i = 0; i < ntri; ++i) parent[i] = (uint32_t) i;
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (const Prim & p : prims) {
        TriGeo & g = pl.geo[p.tri];
        const float * c = out.corners[p.tri].v;
        const v3 p0 = mk3(c[0], c[1], c[2]), e0f = mk3(c[3], c[4], c[5]) - p0, e1f = mk3(c[6], c[7], c[8]) - p0;   // the kernels' float edges
        const double e0[3] = {e0f.x, e0f.y, e0f.z}, e1[3] = {e1f.x, e1f.y, e1f.z};
        const double N[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
        const double nl = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        const double l0 = std::sqrt(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2]), l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
        g.usable = nl > 0.0 && l0 * l1 <= 100.0 * nl;               // slivers (kappa > 100) stay out of every group
        if (!g.usable) continue;
        g.area = 0.5 * nl;
        g.kappa = l0 * l1 / nl;
        g.L = l0 + l1;
        for (int a = 0; a < 3; ++a) {
            g.n[a] = N[a] / nl;
            g.v[0][a] = c[a]; g.v[1][a] = (double) c[a] + e0[a]; g.v[2][a] = (double) c[a] + e1[a];
        }
    }
    auto coplanar = [&](const TriGeo & x, const TriGeo & y) {
        double dm = 0.0, dp = 0.0;
        for (int a = 0; a < 3; ++a) { dm += (x.n[a] - y.n[a]) * (x.n[a] - y.n[a]); dp += (x.n[a] + y.n[a]) * (x.n[a] + y.n[a]); }
        if (std::min(dm, dp) > 1e-8) return false;
        for (int k = 0; k < 3; ++k) {
            double hx = 0.0, hy = 0.0;
            for (int a = 0; a < 3; ++a) { hx += x.n[a] * (y.v[k][a] - x.v[0][a]); hy += y.n[a] * (x.v[k][a] - y.v[0][a]); }
            if (std::fabs(hx) > 5e-6 || std::fabs(hy) > 5e-6) return false;
        }
        return true;
    };
    // incidence by vertex POSITION (meshes that do not share indices still share coordinates along common edges)
    struct Corner { float x, y, z; uint32_t tri; };
    std::vector<Corner> corners;
    corners.reserve(prims.size() * 3);
    for (const Prim & p : prims) {
        if (!pl.geo[p.tri].usable) continue;
        const float * c = out.corners[p.tri].v;
        for (int k = 0; k < 3; ++k) corners.push_back(Corner{c[3 * k], c[3 * k + 1], c[3 * k + 2], p.tri});
    }
    std::sort(corners.begin(), corners.end(), [](const Corner & a, const Corner & b) {
        return std::tie(a.x, a.y, a.z, a.tri) < std::tie(b.x, b.y, b.z, b.tri);
    });
    for (size_t i = 0; i < corners.size();) {
        size_t j = i + 1;
        while (j < corners.size() && corners[j].x == corners[i].x && corners[j].y == corners[i].y && corners[j].z == corners[i].z) ++j;
        if (j - i <= 64)                                      // (a fan of more triangles than that around one point: left alone)
            for (size_t a = i; a < j; ++a)
                for (size_t b2 = a + 1; b2 < j; ++b2) {
                    const uint32_t ra = find(corners[a].tri), rb = find(corners[b2].tri);
                    if (ra != rb && coplanar(pl.geo[corners[a].tri], pl.geo[corners[b2].tri])) parent[ra] = rb;
                }
        i = j;
    }
    std::vector<uint32_t> root_group(ntri, 0xFFFFFFFFu);
    for (const Prim & p : prims) {
        if (!pl.geo[p.tri].usable) continue;
        const uint32_t r = find(p.tri);
        if (root_group[r] == 0xFFFFFFFFu) { root_group[r] = (uint32_t) pl.groups.size(); pl.groups.emplace_back(); }
        pl.gid[p.tri] = root_group[r];
        PlaneGroup & pg = pl.groups[root_group[r]];
        ++pg.size;
        if (pl.geo[p.tri].area > pg.area) {
            pg.area = pl.geo[p.tri].area;
            for (int a = 0; a < 3; ++a) { pg.n[a] = pl.geo[p.tri].n[a]; pg.p[a] = pl.geo[p.tri].v[0][a]; }
        }
    }
    for (const Prim & p : prims) {
        const uint32_t gi = pl.gid[p.tri];
        if (gi == 0xFFFFFFFFu) continue;
        PlaneGroup & pg = pl.groups[gi];
        const TriGeo & g = pl.geo[p.tri];
        for (int k = 0; k < 3; ++k)
            pg.delta = std::max(pg.delta, std::fabs(pg.n[0] * (g.v[k][0] - pg.p[0]) + pg.n[1] * (g.v[k][1] - pg.p[1]) + pg.n[2] * (g.v[k][2] - pg.p[2])));
        // normals as the kernels see them (stored binary32 unit normal) and as the effective triangle has them, up to sign
        const double stored[3] = {out.shade[p.tri].n[0], out.shade[p.tri].n[1], out.shade[p.tri].n[2]};
        for (const double * nn : {stored, g.n}) {
            double dm = 0.0, dpl = 0.0;
            for (int a = 0; a < 3; ++a) { dm += (nn[a] - pg.n[a]) * (nn[a] - pg.n[a]); dpl += (nn[a] + pg.n[a]) * (nn[a] + pg.n[a]); }
            pg.theta = std::max(pg.theta, std::sqrt(std::min(dm, dpl)));
        }
        pg.g = std::max(pg.g, 34.0 * eps * (g.L + 0.02) * g.kappa);
        pg.d = std::max(pg.d, 16.0 * eps * g.kappa);
    }
    for (PlaneGroup & pg : pl.groups)
        pg.ok = pg.delta <= 1e-5 && pg.theta <= 1e-3;
}

// After the 4-wide tree exists: the skip fields of TriShade per triangle (reference of the highest child slot whose triangles all belong to the
// triangle's plane group; a, b of the rule above).
void assign_skips(BuiltScene & out, const Planes & pl, float maxabs)
{
    const double eps = 0x1p-24, EPS = (double) RVB_EPSILON;
    const std::vector<uint32_t> & gid = pl.gid;
    if (gid.empty()) return;                                  // plane analysis switched off
    // per child slot: the group all triangles below share (MIXED = none); children have larger node indices than parents
    const uint32_t MIXED = 0xFFFFFFFFu;
    std::vector<uint32_t> node_group(out.nodes.size() * 4, MIXED);
    auto leaf_group = [&](uint32_t ref) {
        const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
        uint32_t gsel = gid[out.tris[first].index];
        for (uint32_t j = 1; j < count; ++j)
            if (gid[out.tris[first + j].index] != gsel) return MIXED;
        return (gsel != MIXED && pl.groups[gsel].ok) ? gsel : MIXED;
    };
    for (size_t i = out.nodes.size(); i-- > 0;)
        for (int k = 0; k < 4; ++k) {
            const uint32_t ref = out.nodes[i].c[k].ref;
            if (ref == RVB_BVH_EMPTY) continue;
            if (ref & RVB_BVH_LEAF) { node_group[i * 4 + k] = leaf_group(ref); continue; }
            const size_t child = ref >> RVB_BVH_NODE_SHIFT;
            uint32_t gsel = MIXED;
            bool first = true, uniform = true;
            for (int c = 0; c < 4 && uniform; ++c) {
                if (out.nodes[child].c[c].ref == RVB_BVH_EMPTY) continue;
                const uint32_t gg = node_group[child * 4 + c];
                if (gg == MIXED) uniform = false;
                else if (first) { gsel = gg; first = false; }
                else if (gg != gsel) uniform = false;
            }
            node_group[i * 4 + k] = (uniform && !first) ? gsel : MIXED;
        }
    // top-down: a triangle's skip reference is the highest uniform child slot above it
    std::vector<uint32_t> todo(1, 0u), below;
    while (!todo.empty()) {
        const uint32_t i = todo.back();
        todo.pop_back();
        for (int k = 0; k < 4; ++k) {
            const uint32_t ref = out.nodes[i].c[k].ref;
            if (ref == RVB_BVH_EMPTY) continue;
            if (node_group[(size_t) i * 4 + k] == MIXED) {
                if (!(ref & RVB_BVH_LEAF)) todo.push_back(ref >> RVB_BVH_NODE_SHIFT);
                continue;
            }
            below.assign(1, ref);                           // every triangle below this slot skips the slot
            while (!below.empty()) {
                const uint32_t r = below.back();
                below.pop_back();
                if (r & RVB_BVH_LEAF) {
                    const uint32_t first = r & 0x0FFFFFFFu, count = ((r >> 28) & 7u) + 1u;
                    for (uint32_t j = 0; j < count; ++j) out.shade[out.tris[first + j].index].skip_ref = ref;
                } else {
                    for (int c = 0; c < 4; ++c)
                        if (out.nodes[r >> RVB_BVH_NODE_SHIFT].c[c].ref != RVB_BVH_EMPTY) below.push_back(out.nodes[r >> RVB_BVH_NODE_SHIFT].c[c].ref);
                }
            }
        }
    }
    for (const BvhTri & t : out.tris) {
        TriShade & sk = out.shade[t.index];
        sk.group = gid[t.index];
        if (sk.skip_ref == RVB_BVH_EMPTY) continue;
        const PlaneGroup & pg = pl.groups[gid[t.index]];
        const TriGeo & g = pl.geo[t.index];
        const double A = 6.0 * pg.delta + pg.g + 32.0 * eps * g.kappa * g.L + 1.74 * eps * (double) maxabs;
        const double B = 16.0 * eps * g.kappa + 4.0 * eps;
        const double C = 2.0 * pg.theta + pg.d + 8.0 * eps;
        const double a = std::max(0.01, C + A * (1.0 + 1e-4) / EPS), b = B * (1.0 + 1e-4) / EPS;
        sk.skip_a = std::nextafterf((float) a, 2.0f);      // rounded up to binary32
        sk.skip_b = std::nextafterf((float) b, 2.0f);
        if (!(a < 1.0)) sk.skip_ref = RVB_BVH_EMPTY;       // |cos| never exceeds 1: no ray qualifies
    }
}

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
        out.shade[i].skip_ref = RVB_BVH_EMPTY; out.shade[i].skip_a = 2.0f; out.shade[i].skip_b = 0.0f; out.shade[i].group = 0xFFFFFFFFu;
        float c9[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
        std::memcpy(out.corners[i].v, c9, sizeof(c9));
        std::memset(out.corners[i].pad, 0, sizeof(out.corners[i].pad));

        bool finite = true;
        for (float f : c9) finite = finite && std::isfinite(f);
        if (!finite)
            continue;   // NaN/inf vertices: every comparison of the triangle test fails -> never hit
        // |det| <= |d| |e0 x e1|; rvb_set_directions accepts |d| <= 2, so below EPSILON / 2 the triangle can never be hit
        // (quirk Q7).  Evaluated in double with a 1 % margin so nothing hittable is dropped.
        double e0[3] = {(double) p1.x - p0.x, (double) p1.y - p0.y, (double) p1.z - p0.z};
        double e1[3] = {(double) p2.x - p0.x, (double) p2.y - p0.y, (double) p2.z - p0.z};
        double cx = e0[1] * e1[2] - e0[2] * e1[1], cy = e0[2] * e1[0] - e0[0] * e1[2], cz = e0[0] * e1[1] - e0[1] * e1[0];
        if (std::sqrt(cx * cx + cy * cy + cz * cz) * 2.02 < (double) RVB_EPSILON)
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
    Planes planes;
    static const bool plane_skip = !(getenv("RVB_PLANE_SKIP") && getenv("RVB_PLANE_SKIP")[0] == '0');
    // (0 = no units; a unit is larger than a leaf, so a leaf never holds part of one)
    static const uint32_t unit_min_env = getenv("RVB_PLANE_UNIT_MIN") ? (uint32_t) atoi(getenv("RVB_PLANE_UNIT_MIN")) : 8u;
    static const uint32_t unit_min = unit_min_env == 0 ? 0u : std::max(unit_min_env, (uint32_t) RVB_BVH_MAX_LEAF + 1u);
    if (plane_skip && nprims) {
        find_planes(out, b.prims, planes);
        // a plane patch of at least unit_min triangles is built as a subtree of its own
        std::vector<double> centre(planes.groups.size() * 3, 0.0);
        for (const Prim & p : b.prims)
            if (planes.gid[p.tri] != kLoose)
                for (int a = 0; a < 3; ++a) centre[planes.gid[p.tri] * 3 + a] += p.c[a];
        for (Prim & p : b.prims) {
            const uint32_t g = planes.gid[p.tri];
            const bool unit = unit_min && g != kLoose && planes.groups[g].ok && planes.groups[g].size >= unit_min;
            p.unit = unit ? g : kLoose;
            for (int a = 0; a < 3; ++a) p.ukey[a] = unit ? (float) (centre[g * 3 + a] / planes.groups[g].size) : p.c[a];
        }
    } else {
        planes.gid.assign(ntriangles, kLoose);
        for (Prim & p : b.prims)
            for (int a = 0; a < 3; ++a) p.ukey[a] = p.c[a];
    }
    if (nprims == 0) {
        BvhNode root;
        std::memset(&root, 0, sizeof(root));
        for (int c = 0; c < 4; ++c) {
            root.c[c].ref = RVB_BVH_EMPTY;
            root.c[c].lox = root.c[c].loy = root.c[c].loz = 0x7C00u;      // +inf
            root.c[c].hix = root.c[c].hiy = root.c[c].hiz = 0xFC00u;      // -inf
        }
        out.nodes.push_back(root);
        out.depth = 1;
        return "";
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
    b.always_median = attempt == 1;
    for (Prim & p : b.prims)
        for (int a = 0; a < 3; ++a) p.key[a] = p.ukey[a];
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
                    if (!b.nodes[kids[k]].leaf() && !b.nodes[kids[k]].unit_root && b.nodes[kids[k]].box.area() > pick_area) {
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
                // empty slot: an inverted box (lo = +inf, hi = -inf), which no ray enters in the plane-selecting slab test
                // (slab_select: entry = +inf, exit = -inf); the min / max form (slab) sees all of space and rejects the slot by its ref
                slot.lox = slot.loy = slot.loz = 0x7C00u;
                slot.hix = slot.hiy = slot.hiz = 0xFC00u;
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
    if (out.stack_need <= RVB_BVH_STACK) {
        assign_skips(out, planes, maxabs);
        return "";
    }
    }
    return "BVH needs a deeper traversal stack than " + std::to_string(RVB_BVH_STACK) + " entries";
}
