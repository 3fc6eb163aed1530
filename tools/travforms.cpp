// travforms.cpp — developer tool (CPU only; not part of the product, the tests or the bench): replays the path kernel's
// wave-level control flow under alternative traversal FORMS, with the product's own BVH builder (csrc/bvh_build.hip) and
// arithmetic (csrc/rvb_math.h), so that a form can be judged before it is built (round 4; the scheduling POLICIES of one
// form are tools/travsim.cpp's subject).
//
//   travforms <tris.bin> <verts.bin> <dirs.bin> <nrays> <nrefl> sx sy sz          (inputs: tools/dump_scene.py)
//
// A form = rays per wave (lanes per ray) x node width x child order x regrouping interval, priced with wave instructions per
// step kind.  Every form's closest hits are checked against the baseline form's (they must agree: the BVH only prunes).
//
//   pairs        shipped two-lane kernel: 32 rays per wave, 4-wide nodes, nearest child first           (costs 79 / 146 / 75: ISA)
//   pairs-any    ... lowest hit child first (no distance key)                                            (71 / 146 / 75)
//   pairs-w8     ... 8-wide nodes, four children per lane                                                (119 / 146 / 75, estimate)
//   lane1        ONE lane per ray: 64 rays per wave, a lane tests the four children / triangles itself   (costs from the ISA of
//                path_lane_kernel once built; first estimate 90 / 205 / 100)
//   regroup K    rays re-ordered every K bounces by (leaf position of the triangle they stand on, direction octant)
//
// Output: per form, ray-level and wave-level steps per bounce, lanes in the voted step, wave instructions per RAY-bounce.
#include "../parallel-reverb-raytracer_amd/csrc/bvh.h"
#include "../parallel-reverb-raytracer_amd/csrc/rvb_math.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

static std::vector<char> slurp(const char * path)
{
    FILE * f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> b(n);
    if (fread(b.data(), 1, n, f) != (size_t) n) exit(1);
    fclose(f);
    return b;
}
static float half_to_float(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float) h; }
static float clamp_inv(float d) { float inv = 1.0f / d; return inv > 1e30f ? 1e30f : (inv < -1e30f ? -1e30f : inv); }

// ---- a tree of any width derived from the product's 4-wide tree ---------------------------------------------------------------
struct WChild { float lo[3], hi[3]; uint32_t ref; };      // ref: LEAF | ... as in bvh.h, or index of a WNode
struct WNode { std::vector<WChild> c; };
struct WTree { std::vector<WNode> nodes; int width = 4; };

static WChild wchild_of(const BvhChild & ch)
{
    WChild w;
    w.lo[0] = half_to_float(ch.lox); w.lo[1] = half_to_float(ch.loy); w.lo[2] = half_to_float(ch.loz);
    w.hi[0] = half_to_float(ch.hix); w.hi[1] = half_to_float(ch.hiy); w.hi[2] = half_to_float(ch.hiz);
    w.ref = ch.ref;
    return w;
}
static float area(const WChild & c)
{
    const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
// width 4: the product's tree as it stands (refs stay byte offsets >> 6 = node index).  width 8: a node's inner children are opened,
// largest surface area first, while the node then holds at most 8 children (the usual wide-BVH collapse); own-plane skip
// references are kept valid only where the skipped child survives as a child (otherwise the skip is lost: conservative).
static WTree make_tree(const BuiltScene & bs, int width, std::vector<uint32_t> & node_of_ref)
{
    WTree t;
    t.width = width;
    node_of_ref.assign(bs.nodes.size(), 0xFFFFFFFFu);
    std::vector<uint32_t> todo = {0};
    node_of_ref[0] = 0;
    t.nodes.emplace_back();
    for (size_t at = 0; at < todo.size(); ++at) {
        std::vector<WChild> kids;
        for (int k = 0; k < 4; ++k)
            if (bs.nodes[todo[at]].c[k].ref != RVB_BVH_EMPTY) kids.push_back(wchild_of(bs.nodes[todo[at]].c[k]));
        while (width > 4) {
            int best = -1;
            for (size_t i = 0; i < kids.size(); ++i) {
                if (kids[i].ref & RVB_BVH_LEAF) continue;
                int nk = 0;
                for (int k = 0; k < 4; ++k) nk += bs.nodes[kids[i].ref >> RVB_BVH_NODE_SHIFT].c[k].ref != RVB_BVH_EMPTY;
                if ((int) kids.size() - 1 + nk > width) continue;
                if (best < 0 || area(kids[i]) > area(kids[best])) best = (int) i;
            }
            if (best < 0) break;
            const BvhNode & n = bs.nodes[kids[best].ref >> RVB_BVH_NODE_SHIFT];
            kids.erase(kids.begin() + best);
            for (int k = 0; k < 4; ++k) if (n.c[k].ref != RVB_BVH_EMPTY) kids.push_back(wchild_of(n.c[k]));
        }
        for (WChild & c : kids)
            if (!(c.ref & RVB_BVH_LEAF)) {
                const uint32_t old = c.ref >> RVB_BVH_NODE_SHIFT;
                node_of_ref[old] = (uint32_t) t.nodes.size();
                todo.push_back(old);
                t.nodes.emplace_back();
            }
        t.nodes[at].c = kids;            // child refs still name OLD nodes (byte offsets); resolved through node_of_ref at visit time
    }
    return t;
}

struct Query {
    v3 o, d;
    float ix, iy, iz, oix, oiy, oiz, best_t;
    uint32_t best_i, ref, skip;
    std::vector<uint32_t> stack;
    std::vector<uint32_t> stack_b;       // split stacks (pairs): the entries pushed by the ray's second lane (children 2, 3)
    uint64_t stamp_a = 0, stamp_b = 0, clock = 0;
};

struct Form {
    std::string name;
    int rays_per_wave;
    int width;
    bool sorted;
    int regroup;                         // bounces between re-orderings of the rays, 0 = never
    double c_node, c_leaf, c_done;
    int split = 0;                       // stack policy (Sim::split)
    int signed_keys = 0;                 // child keys from the UNCLAMPED entry distance, compared as signed integers (Sim::signed_keys)
};

struct Sim {
    const BuiltScene & bs;
    WTree tree;
    std::vector<uint32_t> node_of_ref;
    float cull_abs, cull_rel;
    bool sorted;
    int split = 0;                       // 0: one stack per ray; 1: a stack per lane of the pair, lane 0's popped first; 2: the larger one first; 3: the one pushed to last first
    int signed_keys = 0;                 // 1: no max(tn, 0) before the key; negative entry distances order by signed comparison (all before the positive ones)
    unsigned long long node_steps = 0, leaf_steps = 0;

    void begin(Query & q, v3 o, v3 d, uint32_t skip)
    {
        q.o = o; q.d = d;
        q.ix = clamp_inv(d.x); q.iy = clamp_inv(d.y); q.iz = clamp_inv(d.z);
        q.oix = o.x * q.ix; q.oiy = o.y * q.iy; q.oiz = o.z * q.iz;
        q.best_t = __builtin_inff(); q.best_i = 0xFFFFFFFFu; q.ref = 0; q.skip = skip;
        q.stack.clear();
        q.stack_b.clear();
    }
    void pop(Query & q)
    {
        if (!split) { if (q.stack.empty()) q.ref = 0xFFFFFFFFu; else { q.ref = q.stack.back(); q.stack.pop_back(); } return; }
        const bool a = !q.stack.empty(), b = !q.stack_b.empty();
        if (!a && !b) { q.ref = 0xFFFFFFFFu; return; }
        bool take_a = a;
        if (a && b) take_a = split == 1 ? true : (split == 2 ? q.stack.size() >= q.stack_b.size() : q.stamp_a >= q.stamp_b);
        std::vector<uint32_t> & st = take_a ? q.stack : q.stack_b;
        q.ref = st.back();
        st.pop_back();
    }
    void node_step(Query & q)
    {
        ++node_steps;
        const WNode & n = tree.nodes[node_of_ref[q.ref >> RVB_BVH_NODE_SHIFT]];
        const float lim = fmaf(q.best_t, 1.0f + cull_rel, cull_abs), neg_cull = -cull_abs;
        int nok = 0, okc[16]; float tn[16];
        for (size_t c = 0; c < n.c.size(); ++c) {
            const WChild & ch = n.c[c];
            const float tx0 = fmaf(ch.lo[0], q.ix, -q.oix), tx1 = fmaf(ch.hi[0], q.ix, -q.oix);
            const float ty0 = fmaf(ch.lo[1], q.iy, -q.oiy), ty1 = fmaf(ch.hi[1], q.iy, -q.oiy);
            const float tz0 = fmaf(ch.lo[2], q.iz, -q.oiz), tz1 = fmaf(ch.hi[2], q.iz, -q.oiz);
            const float a = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), neg_cull));
            const float b = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), lim));
            if (a <= b && ch.ref != q.skip) { okc[nok] = (int) c; tn[nok] = signed_keys ? a : fmaxf(a, 0.0f); ++nok; }
        }
        if (!nok) { pop(q); return; }
        int w = 0;
        if (sorted) {
            uint32_t bestk = 0xFFFFFFFFu;
            int32_t bests = 0x7FFFFFFF;
            for (int i = 0; i < nok; ++i) {
                uint32_t fb; memcpy(&fb, &tn[i], 4);
                const uint32_t k = (fb & ~(tree.width > 4 ? 7u : 3u)) | (uint32_t) okc[i];
                if (signed_keys) { if ((int32_t) k < bests) { bests = (int32_t) k; w = i; } }
                else if (k < bestk) { bestk = k; w = i; }
            }
        }
        for (int i = 0; i < nok; ++i) if (i != w) {
            if (split && okc[i] >= 2) { q.stack_b.push_back(n.c[okc[i]].ref); q.stamp_b = ++q.clock; }
            else { q.stack.push_back(n.c[okc[i]].ref); q.stamp_a = ++q.clock; }
        }
        q.ref = n.c[okc[w]].ref;
    }
    void leaf_step(Query & q)
    {
        ++leaf_steps;
        const uint32_t first = q.ref & 0x0FFFFFFFu, count = ((q.ref >> 28) & 7u) + 1u;
        for (uint32_t j = 0; j < count; ++j) {
            const BvhTri & t = bs.tris[first + j];
            const float dist = mt_intersect(mk3(t.v0[0], t.v0[1], t.v0[2]), mk3(t.e0[0], t.e0[1], t.e0[2]), mk3(t.e1[0], t.e1[1], t.e1[2]), q.o, q.d);
            if (dist > RVB_EPSILON && (q.best_i == 0xFFFFFFFFu || dist < q.best_t || (dist == q.best_t && t.index < q.best_i))) { q.best_t = dist; q.best_i = t.index; }
        }
        pop(q);
    }
};

struct RayState { v3 o, d; uint32_t bounce, skip, tri; bool alive; uint32_t id; };

struct Result { double ray_node, ray_leaf, wave_node, wave_leaf, wave_done, act_node, act_leaf, act_done, cost; unsigned long long bounces; unsigned long long checksum; };

// one wave runs its rays for up to `bounces_here` bounces each (vote loop of traverse_pairs_vote)
static unsigned long long g_votes = 0;
static const bool g_chain = getenv("TRAVFORMS_CHAIN") != nullptr;
static const int g_double = getenv("TRAVFORMS_DOUBLE") ? atoi(getenv("TRAVFORMS_DOUBLE")) : 0;
static void run_wave(Sim & s, std::vector<RayState *> & rays, uint32_t stop_bounce, uint32_t nrefl, unsigned long long cnt[3], unsigned long long act[3],
                     unsigned long long & bounces, unsigned long long & checksum)
{
    const int nq = (int) rays.size();
    enum St { NODE, LEAF, DONE, IDLE };
    std::vector<Query> q(nq);
    std::vector<St> st(nq, IDLE);
    for (int i = 0; i < nq; ++i)
        if (rays[i]->alive && rays[i]->bounce < stop_bounce) { s.begin(q[i], rays[i]->o, rays[i]->d, rays[i]->skip); st[i] = NODE; }
    auto classify = [&](int i) { st[i] = q[i].ref == 0xFFFFFFFFu ? DONE : ((q[i].ref & RVB_BVH_LEAF) ? LEAF : NODE); };
    auto do_step = [&](int a) {
        for (int i = 0; i < nq; ++i) {
            if (st[i] != (St) a) continue;
            if (a == NODE) { s.node_step(q[i]); classify(i); }
            else if (a == LEAF) { s.leaf_step(q[i]); classify(i); }
            else {
                RayState & r = *rays[i];
                if (q[i].best_i == 0xFFFFFFFFu) { r.alive = false; st[i] = IDLE; continue; }
                ++bounces;
                uint32_t tb; memcpy(&tb, &q[i].best_t, 4);
                checksum += (unsigned long long) (q[i].best_i + 1) * 0x9E3779B97F4A7C15ull + tb * (unsigned long long) (r.id * 131u + r.bounce + 7u);
                const TriShade & sh = s.bs.shade[q[i].best_i];
                const v3 n = mk3(sh.n[0], sh.n[1], sh.n[2]);
                const v3 pnt = r.o + r.d * q[i].best_t;
                const float thr = fmaf(sh.skip_b, q[i].best_t, sh.skip_a), cosine = fabsf(dot3(n, r.d));
                r.d = reflect3(n, r.d);
                r.o = pnt;
                r.tri = q[i].best_i;
                r.skip = cosine > thr ? sh.skip_ref : RVB_BVH_EMPTY;
                if (s.tree.width > 4 && r.skip != RVB_BVH_EMPTY) {
                    // the skipped child must still be a child somewhere in the widened tree (an opened node is no child any more)
                    bool kept = (r.skip & RVB_BVH_LEAF) != 0;
                    if (!kept) kept = s.node_of_ref[r.skip >> RVB_BVH_NODE_SHIFT] != 0xFFFFFFFFu;
                    if (!kept) r.skip = RVB_BVH_EMPTY;
                }
                ++r.bounce;
                if (r.bounce >= nrefl) { r.alive = false; st[i] = IDLE; continue; }
                if (r.bounce >= stop_bounce) { st[i] = IDLE; continue; }
                s.begin(q[i], r.o, r.d, r.skip);
                st[i] = NODE;
            }
        }
    };
    const char * cyc = getenv("TRAVFORMS_CYCLE");        // "TL,TD": no vote — every iteration a node step, then a leaf step if TL % of the live lanes
    int TL = 0, TD = 0;                                   // wait for one, then a shading step if TD % do (or if nothing else can run)
    if (cyc) sscanf(cyc, "%d,%d", &TL, &TD);
    for (;;) {
        int c[3] = {0, 0, 0};
        for (int i = 0; i < nq; ++i) if (st[i] != IDLE) ++c[st[i]];
        if (c[0] + c[1] + c[2] == 0) break;
        if (cyc) {
            ++g_votes;
            const int live = c[0] + c[1] + c[2];
            bool ran = false;
            static const int node_reps = getenv("TRAVFORMS_CYCLE_NODES") ? atoi(getenv("TRAVFORMS_CYCLE_NODES")) : 1;
            static const int TN = getenv("TRAVFORMS_CYCLE_TN") ? atoi(getenv("TRAVFORMS_CYCLE_TN")) : 0;     // node step only if TN % of the live lanes are at a node (or nothing else would run)
            for (int rep = 0; rep < node_reps; ++rep) {
                int at = 0, lf = 0, dd = 0;
                for (int i = 0; i < nq; ++i) { at += st[i] == NODE; lf += st[i] == LEAF; dd += st[i] == DONE; }
                const bool others = (lf && lf * 100 >= TL * live) || (dd && dd * 100 >= TD * live);
                if (at && (at * 100 >= TN * live || !others)) { ++cnt[NODE]; act[NODE] += at; do_step(NODE); ran = true; }
            }
            int l = 0, dn = 0;
            for (int i = 0; i < nq; ++i) { l += st[i] == LEAF; dn += st[i] == DONE; }
            if (l && (l * 100 >= TL * live || !ran)) { ++cnt[LEAF]; act[LEAF] += l; do_step(LEAF); ran = true; }
            dn = 0;
            for (int i = 0; i < nq; ++i) dn += st[i] == DONE;
            if (dn && (dn * 100 >= TD * live || !ran)) { ++cnt[DONE]; act[DONE] += dn; do_step(DONE); }
            continue;
        }
        int a = NODE;                                            // n_node >= n_leaf && n_node >= n_done, else leaf >= done, else done
        if (!(c[0] >= c[1] && c[0] >= c[2])) a = c[1] >= c[2] ? LEAF : DONE;
        ++cnt[a]; act[a] += c[a];
        ++g_votes;
        const bool strong = a == NODE && g_double > 0 && c[0] * 100 >= g_double * (c[0] + c[1] + c[2]);
        do_step(a);
        if ((g_chain && a != NODE) || strong) {
            // TRAVFORMS_CHAIN=1: a leaf or shading step is followed by a node step for the lanes that are at a node by then, without a vote;
            // TRAVFORMS_DOUBLE=P: a node step voted for by P % of the live lanes or more is followed by a second one without a vote
            int at_node = 0;
            for (int i = 0; i < nq; ++i) at_node += st[i] == NODE;
            if (at_node) { ++cnt[NODE]; act[NODE] += at_node; do_step(NODE); }
        }
    }
}

static Result run_form(const BuiltScene & bs, const Form & f, const float * dirs, uint64_t nrays, uint32_t nrefl, v3 src)
{
    Sim s{bs};
    s.tree = make_tree(bs, f.width, s.node_of_ref);
    s.cull_abs = bs.pad; s.cull_rel = 1e-4f; s.sorted = f.sorted; s.split = f.split; s.signed_keys = f.signed_keys;
    std::vector<RayState> rays(nrays);
    for (uint64_t i = 0; i < nrays; ++i) rays[i] = RayState{src, mk3(dirs[4 * i], dirs[4 * i + 1], dirs[4 * i + 2]), 0, RVB_BVH_EMPTY, 0xFFFFFFFFu, true, (uint32_t) i};
    std::vector<uint32_t> order(nrays);
    std::iota(order.begin(), order.end(), 0u);
    unsigned long long cnt[3] = {0, 0, 0}, act[3] = {0, 0, 0}, bounces = 0, checksum = 0;
    const uint32_t slab = f.regroup ? (uint32_t) f.regroup : nrefl;
    for (uint32_t b0 = 0; b0 < nrefl; b0 += slab) {
        if (f.regroup && b0) {
            // key: leaf position of the triangle the ray stands on (spatially coherent order of the builder), then direction octant;
            // dead rays last
            std::vector<uint64_t> key(nrays);
            for (uint64_t i = 0; i < nrays; ++i) {
                const RayState & r = rays[i];
                if (!r.alive) { key[i] = ~0ull; continue; }
                const uint32_t oct = (r.d.x < 0 ? 1u : 0u) | (r.d.y < 0 ? 2u : 0u) | (r.d.z < 0 ? 4u : 0u);
                const uint32_t lp = bs.leafpos[r.tri];
                const int shift = getenv("TRAVFORMS_KEY_SHIFT") ? atoi(getenv("TRAVFORMS_KEY_SHIFT")) : 6;
                const bool oct_first = getenv("TRAVFORMS_OCT_FIRST") != nullptr;
                key[i] = oct_first ? (((uint64_t) oct << 40) | lp) : ((((uint64_t) (lp >> shift)) << 3 | oct) << 32 | lp);
            }
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        }
        for (uint64_t w = 0; w < nrays; w += f.rays_per_wave) {
            std::vector<RayState *> wave;
            for (uint64_t i = w; i < std::min<uint64_t>(nrays, w + f.rays_per_wave); ++i) wave.push_back(&rays[order[i]]);
            run_wave(s, wave, std::min(nrefl, b0 + slab), nrefl, cnt, act, bounces, checksum);
        }
    }
    Result r;
    const double R = f.rays_per_wave;
    r.bounces = bounces; r.checksum = checksum;
    r.ray_node = (double) s.node_steps / bounces; r.ray_leaf = (double) s.leaf_steps / bounces;
    r.wave_node = R * cnt[0] / bounces; r.wave_leaf = R * cnt[1] / bounces; r.wave_done = R * cnt[2] / bounces;
    r.act_node = (double) act[0] / cnt[0]; r.act_leaf = (double) act[1] / cnt[1]; r.act_done = (double) act[2] / cnt[2];
    r.cost = (r.wave_node * f.c_node + r.wave_leaf * f.c_leaf + r.wave_done * f.c_done) / R;
    return r;
}

int main(int argc, char ** argv)
{
    if (argc < 9) { fprintf(stderr, "usage: travforms tris.bin verts.bin dirs.bin nrays nrefl sx sy sz [form ...]\n"); return 1; }
    auto tb = slurp(argv[1]), vb = slurp(argv[2]), db = slurp(argv[3]);
    const uint64_t ntri = tb.size() / 32, nvert = vb.size() / 16;
    uint64_t nrays = strtoull(argv[4], 0, 10);
    const uint32_t nrefl = atoi(argv[5]);
    if (nrays > db.size() / 16) nrays = db.size() / 16;
    const v3 src = mk3(atof(argv[6]), atof(argv[7]), atof(argv[8]));
    BuiltScene bs;
    const std::string err = rvb_build_scene((const rvb_triangle *) tb.data(), ntri, (const rvb_float3 *) vb.data(), nvert, 1000, bs);
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    printf("scene: %zu triangles kept, %zu nodes (4-wide), depth %u; %llu rays x %u bounces\n", bs.tris.size(), bs.nodes.size(), bs.depth,
           (unsigned long long) nrays, nrefl);
    // wave instructions per step kind (VALU, incl. the vote): pairs from profiles/r03_isa_mix_path_pair_kernel.txt; lane1 from
    // TRAVFORMS_LANE1_COSTS (the ISA of path_lane_kernel) or the first estimate
    double l1n = 90, l1l = 205, l1d = 100;
    if (getenv("TRAVFORMS_LANE1_COSTS")) sscanf(getenv("TRAVFORMS_LANE1_COSTS"), "%lf,%lf,%lf", &l1n, &l1l, &l1d);
    std::vector<Form> forms = {
        {"pairs (shipped)", 32, 4, true, 0, 79, 146, 75},
        {"pairs, lowest hit child first", 32, 4, false, 0, 71, 146, 75},
        {"pairs, 8-wide nodes", 32, 8, true, 0, 119, 146, 75},
        {"pairs, 8-wide, lowest child first", 32, 8, false, 0, 107, 146, 75},
        {"one lane per ray", 64, 4, true, 0, l1n, l1l, l1d},
        {"one lane per ray, lowest child first", 64, 4, false, 0, l1n - 10, l1l, l1d},
    };
    // a stack per LANE of the pair (pushes need no mask exchange: 12 vector instructions fewer per node step); which stack is popped first
    forms.push_back({"pairs, stack per lane, lane 0's first", 32, 4, true, 0, 67, 146, 75, 1});
    forms.push_back({"pairs, stack per lane, larger first", 32, 4, true, 0, 67, 146, 75, 2});
    forms.push_back({"pairs, stack per lane, last pushed first", 32, 4, true, 0, 67, 146, 75, 3});
    forms.push_back({"quads (four lanes per ray, 16 rays per wave)", 16, 4, true, 0, 51, 89, 85});
    forms.push_back({"pairs, pushes from counts (round 4: 61 / 143 / 98 + 4 copies per step)", 32, 4, true, 0, 65, 147, 102, 0, 0});
    forms.push_back({"pairs, pushes from counts, signed keys of the unclamped entry distance", 32, 4, true, 0, 63, 147, 102, 0, 1});
    for (int K : {1, 2, 4, 8, 16, 32}) {
        forms.push_back({"pairs, regroup every " + std::to_string(K), 32, 4, true, K, 79, 146, 75});
        forms.push_back({"one lane per ray, regroup every " + std::to_string(K), 64, 4, true, K, l1n, l1l, l1d});
    }
    const char * only = getenv("TRAVFORMS_ONLY");
    double base_cost = 0;
    unsigned long long base_sum = 0, base_bounces = 0;
    for (size_t i = 0; i < forms.size(); ++i) {
        const Form & f = forms[i];
        if (only && i && f.name.find(only) == std::string::npos) continue;
        g_votes = 0;
        const Result r = run_form(bs, f, (const float *) db.data(), nrays, nrefl, src);
        if (i == 0) { base_cost = r.cost; base_sum = r.checksum; base_bounces = r.bounces; }
        printf("%-44s | ray node %5.2f leaf %4.2f | wave node %5.2f leaf %5.2f done %4.2f per %2d ray-bounces | lanes in step: node %.2f leaf %.2f done %.2f | "
               "wave instructions per ray-bounce %6.1f (%+5.1f %%)%s\n",
               f.name.c_str(), r.ray_node, r.ray_leaf, r.wave_node, r.wave_leaf, r.wave_done, f.rays_per_wave,
               r.act_node / f.rays_per_wave, r.act_leaf / f.rays_per_wave, r.act_done / f.rays_per_wave, r.cost, 100.0 * (r.cost / base_cost - 1.0),
               (r.checksum == base_sum && r.bounces == base_bounces) ? "" : "  ** HITS DIFFER FROM THE BASELINE FORM **");
        printf("%-44s | votes %5.2f per %d ray-bounces%s\n", "", (double) g_votes * f.rays_per_wave / (double) r.bounces, f.rays_per_wave, g_chain ? " (TRAVFORMS_CHAIN: node step behind every leaf / shading step)" : "");
        fflush(stdout);
    }
    return 0;
}
