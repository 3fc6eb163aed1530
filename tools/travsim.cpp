// travsim.cpp — developer tool (CPU only, not part of the product or the tests): replays the quad
// traversal of csrc/trace_kernels.hip on the host for a dumped scene and counts ray-level and
// wave-level steps under alternative traversal policies, so that a policy can be judged before a
// GPU run is spent on it.  Uses the product's own BVH builder and triangle arithmetic.
//
//   travsim <tris.bin> <verts.bin> <dirs.bin> <nrays> <nrefl> sx sy sz mx my mz
#include "../parallel-reverb-raytracer_amd/csrc/bvh.h"
#include "../parallel-reverb-raytracer_amd/csrc/rvb_math.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
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

static float half_to_float(uint16_t b)
{
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float) h;
}

struct Policy {
    bool stack_dist = false;     // stack entries carry the entry distance; culled at pop time without a node visit
    bool sorted_push = false;    // siblings pushed far-to-near (nearest popped first) instead of lane order
    int postpone = 0;
    int any_order = 0;           // any-hit child order: 0 first hit lane (shipped), 1 nearest first, 2 farthest first            // leaves a quad may hold back while it keeps walking nodes (Aila-Laine speculative traversal)
};

struct Query {
    v3 o, d;
    float ix, iy, iz, oix, oiy, oiz;
    float best_t, tmax;
    uint32_t best_i;
    bool any;
    uint32_t ref;
    std::vector<uint32_t> stack;
    std::vector<float> stack_t;
    bool found = false;
    std::vector<uint32_t> held;  // postponed leaves
    uint32_t skip = 0xFFFFFFFFu; // subtree that cannot hold a hit of this query (BuiltScene::skip_ref)
};

static float clamp_inv(float d)
{
    float inv = 1.0f / d;
    if (inv > 1e30f) inv = 1e30f;
    if (inv < -1e30f) inv = -1e30f;
    return inv;
}

struct Sim {
    BuiltScene bs;
    float cull_abs, cull_rel;
    Policy pol;
    // counters
    unsigned long long node_steps = 0, node_steps_allculled = 0, leaf_steps = 0, pop_culls = 0, pushes = 0;
    unsigned long long by_level[32] = {0}, hits_by_level[32] = {0};
    std::vector<uint8_t> level;          // depth of every node

    void begin(Query & q, v3 o, v3 d, bool any, float tmax)
    {
        q.o = o; q.d = d; q.any = any; q.tmax = tmax;
        q.ix = clamp_inv(d.x); q.iy = clamp_inv(d.y); q.iz = clamp_inv(d.z);
        q.oix = o.x * q.ix; q.oiy = o.y * q.iy; q.oiz = o.z * q.iz;
        q.best_t = any ? tmax : __builtin_inff();
        q.best_i = 0xFFFFFFFFu;
        q.ref = 0;
        q.stack.clear(); q.stack_t.clear();
        q.found = false;
        q.held.clear();
        q.skip = 0xFFFFFFFFu;
    }
    float limit(const Query & q) const { return fmaf(q.best_t, 1.0f + cull_rel, cull_abs); }
    void pop(Query & q)
    {
        for (;;) {
            if (q.stack.empty()) { q.ref = 0xFFFFFFFFu; return; }
            q.ref = q.stack.back();
            float t = q.stack_t.back();
            q.stack.pop_back(); q.stack_t.pop_back();
            if (pol.stack_dist && t > limit(q)) { ++pop_culls; continue; }
            return;
        }
    }
    // one node step; q.ref must be a node
    void node_step(Query & q)
    {
        ++node_steps;
        const BvhNode & n = bs.nodes[q.ref >> RVB_BVH_NODE_SHIFT];
        const int lvl = level.empty() ? 0 : level[q.ref >> RVB_BVH_NODE_SHIFT];
        ++by_level[lvl];
        const float lim = limit(q), neg_cull = getenv("TRAVSIM_ZERO_CLAMP") ? 0.0f : -cull_abs;
        bool ok[4]; float tn[4];
        int nok = 0;
        for (int c = 0; c < 4; ++c) {
            const BvhChild & ch = n.c[c];
            float lox = half_to_float(ch.lox), loy = half_to_float(ch.loy), loz = half_to_float(ch.loz);
            float hix = half_to_float(ch.hix), hiy = half_to_float(ch.hiy), hiz = half_to_float(ch.hiz);
            float tx0 = fmaf(lox, q.ix, -q.oix), tx1 = fmaf(hix, q.ix, -q.oix);
            float ty0 = fmaf(loy, q.iy, -q.oiy), ty1 = fmaf(hiy, q.iy, -q.oiy);
            float tz0 = fmaf(loz, q.iz, -q.oiz), tz1 = fmaf(hiz, q.iz, -q.oiz);
            float a = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), neg_cull));
            float b = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), lim));
            ok[c] = a <= b && ch.ref != RVB_BVH_EMPTY && ch.ref != q.skip;
            tn[c] = fmaxf(a, 0.0f);
            nok += ok[c];
        }
        hits_by_level[lvl] += nok;
        if (!nok) { ++node_steps_allculled; pop(q); return; }
        int winner = -1;
        if (q.any && pol.any_order == 0) { for (int c = 0; c < 4; ++c) if (ok[c]) { winner = c; break; } }
        else if (q.any) { for (int c = 0; c < 4; ++c) if (ok[c] && (winner < 0 || (pol.any_order == 1 ? tn[c] < tn[winner] : tn[c] > tn[winner]))) winner = c; }
        else {
            uint32_t bestk = 0xFFFFFFFFu;
            for (int c = 0; c < 4; ++c) if (ok[c]) {
                uint32_t fb; memcpy(&fb, &tn[c], 4);
                uint32_t k = (fb & ~3u) | c;
                if (k < bestk) { bestk = k; winner = c; }
            }
        }
        int order[4], m = 0;
        for (int c = 0; c < 4; ++c) if (ok[c] && c != winner) order[m++] = c;
        if (pol.sorted_push && !q.any) {                       // far first
            for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) if (tn[order[j]] > tn[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
        }
        for (int i = 0; i < m; ++i) { q.stack.push_back(n.c[order[i]].ref); q.stack_t.push_back(tn[order[i]]); ++pushes; }
        q.ref = n.c[winner].ref;
    }
    // the triangle tests of one leaf, no stack movement
    void leaf_test(Query & q, uint32_t ref)
    {
        ++leaf_steps;
        const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
        for (uint32_t j = 0; j < count; ++j) {
            const BvhTri & t = bs.tris[first + j];
            float dist = mt_intersect(mk3(t.v0[0], t.v0[1], t.v0[2]), mk3(t.e0[0], t.e0[1], t.e0[2]), mk3(t.e1[0], t.e1[1], t.e1[2]), q.o, q.d);
            if (dist > RVB_EPSILON && (q.best_i == 0xFFFFFFFFu || dist < q.best_t || (dist == q.best_t && t.index < q.best_i))) { q.best_t = dist; q.best_i = t.index; }
        }
    }
    // one leaf step; returns true when the query has finished
    bool leaf_step(Query & q)
    {
        ++leaf_steps;
        const uint32_t first = q.ref & 0x0FFFFFFFu, count = ((q.ref >> 28) & 7u) + 1u;
        for (uint32_t j = 0; j < count; ++j) {
            const BvhTri & t = bs.tris[first + j];
            float dist = mt_intersect(mk3(t.v0[0], t.v0[1], t.v0[2]), mk3(t.e0[0], t.e0[1], t.e0[2]), mk3(t.e1[0], t.e1[1], t.e1[2]), q.o, q.d);
            if (q.any) { if (dist > RVB_EPSILON && dist <= q.tmax) q.found = true; }
            else if (dist > RVB_EPSILON && (q.best_i == 0xFFFFFFFFu || dist < q.best_t || (dist == q.best_t && t.index < q.best_i))) { q.best_t = dist; q.best_i = t.index; }
        }
        if (q.found) return true;
        pop(q);
        return q.ref == 0xFFFFFFFFu;
    }
};

struct Ray { v3 o, d; uint32_t bounce; bool alive; };

int main(int argc, char ** argv)
{
    if (argc < 12) { fprintf(stderr, "usage\n"); return 1; }
    auto tb = slurp(argv[1]), vb = slurp(argv[2]), db = slurp(argv[3]);
    const uint64_t ntri = tb.size() / 32, nvert = vb.size() / 16;
    uint64_t nrays = strtoull(argv[4], 0, 10);
    const uint32_t nrefl = atoi(argv[5]);
    if (nrays > db.size() / 16) nrays = db.size() / 16;
    v3 src = mk3(atof(argv[6]), atof(argv[7]), atof(argv[8])), mic = mk3(atof(argv[9]), atof(argv[10]), atof(argv[11]));
    Sim base;
    std::string err = rvb_build_scene((const rvb_triangle *) tb.data(), ntri, (const rvb_float3 *) vb.data(), nvert, 1000, base.bs);
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    base.cull_abs = base.bs.pad; base.cull_rel = 1e-4f;
    base.level.assign(base.bs.nodes.size(), 0);
    for (size_t i = 0; i < base.bs.nodes.size(); ++i)
        for (int k = 0; k < 4; ++k) {
            const uint32_t r = base.bs.nodes[i].c[k].ref;
            if (r != RVB_BVH_EMPTY && !(r & RVB_BVH_LEAF)) base.level[r >> RVB_BVH_NODE_SHIFT] = base.level[i] + 1;
        }
    {
        size_t leaves = 0, hist[5] = {0, 0, 0, 0, 0}, kids[5] = {0, 0, 0, 0, 0};
        for (const BvhNode & n : base.bs.nodes) {
            int nk = 0;
            for (int k = 0; k < 4; ++k) {
                const uint32_t r = n.c[k].ref;
                if (r == RVB_BVH_EMPTY) continue;
                ++nk;
                if (r & RVB_BVH_LEAF) { ++leaves; ++hist[((r >> 28) & 7u) + 1]; }
            }
            ++kids[nk];
        }
        printf("nodes %zu tris %zu depth %u stack_need %u | leaves %zu with 1/2/3/4 tris: %zu %zu %zu %zu | nodes with 1/2/3/4 children: %zu %zu %zu %zu\n",
               base.bs.nodes.size(), base.bs.tris.size(), base.bs.depth, base.bs.stack_need, leaves, hist[1], hist[2], hist[3], hist[4], kids[1], kids[2], kids[3], kids[4]);
    }
    {
        // own-plane skip statistics: triangles by the tree level of the node that holds their skip child
        std::vector<int> slot_level(base.bs.nodes.size() * 4 + 4, -1);
        size_t by_lvl[16] = {0}, none = 0, ngroups = 0;
        std::vector<uint32_t> parent_level_of_ref;
        for (const BvhTri & t : base.bs.tris) {
            const TriShade & sk = base.bs.shade[t.index];
            if (sk.group != 0xFFFFFFFFu && sk.group + 1 > ngroups) ngroups = sk.group + 1;
            if (sk.skip_ref == RVB_BVH_EMPTY) { ++none; continue; }
            int lvl = 15;
            if (!(sk.skip_ref & RVB_BVH_LEAF)) lvl = base.level[sk.skip_ref >> RVB_BVH_NODE_SHIFT] - 1;
            else lvl = 14;                                // leaf-level skip
            ++by_lvl[lvl < 0 ? 0 : lvl];
        }
        printf("own-plane skip: %zu plane groups; triangles without a skip subtree %zu; by level of the skipped child's parent:", ngroups, none);
        for (int l = 0; l < 14; ++l) if (by_lvl[l]) printf(" L%d %zu", l, by_lvl[l]);
        printf(" leaf-only %zu\n", by_lvl[14]);
    }
    const float * dirs = (const float *) db.data();
    const bool verify = getenv("TRAVSIM_VERIFY") != nullptr;
    const double grazing = getenv("TRAVSIM_GRAZING") ? atof(getenv("TRAVSIM_GRAZING")) : 0.0;
    std::vector<float> gdirs;
    if (grazing > 0.0) {                              // keep azimuth, squash the elevation: |d.y| in (0, grazing]
        gdirs.assign(dirs, dirs + 4 * nrays);
        for (uint64_t i = 0; i < nrays; ++i) {
            double x = gdirs[4 * i], y = gdirs[4 * i + 1], z = gdirs[4 * i + 2];
            double h = std::sqrt(x * x + z * z);
            if (h == 0) { x = 1; h = 1; }
            double e = grazing * (0.001 + 0.999 * std::fabs(y)) * (y < 0 ? -1 : 1);
            double n = std::sqrt(1 + e * e);
            gdirs[4 * i] = (float) (x / h / n); gdirs[4 * i + 1] = (float) (e / n); gdirs[4 * i + 2] = (float) (z / h / n);
        }
        dirs = gdirs.data();
    }
    unsigned long long verified = 0, mismatches = 0, verified_any = 0, mismatches_any = 0;

    const int NQ = getenv("TRAVSIM_RAYS_PER_WAVE") ? atoi(getenv("TRAVSIM_RAYS_PER_WAVE")) : 16;   // rays that vote together (first replay only)
    // wave instructions per step (from the ISA, incl. ~6 for the vote): quad kernel by default, TRAVSIM_COSTS="79,146,75" = pair kernel
    double C_NODE = 56, C_LEAF = 125, C_DONE = 115;
    if (getenv("TRAVSIM_COSTS")) sscanf(getenv("TRAVSIM_COSTS"), "%lf,%lf,%lf", &C_NODE, &C_LEAF, &C_DONE);
    for (int p = 0; p < (verify ? 1 : 14); ++p) {
        Sim s = base;
        s.pol.stack_dist = 0; s.pol.sorted_push = 0; s.pol.postpone = 0;
        // scheduler: 0 = while-while (shipped); 1 = majority vote; 2..: node loop runs while >= T quads want a node step
        const int sched = p;
        const int T = p >= 2 ? (p - 1) * 2 : 1;
        unsigned long long w_node = 0, w_leaf = 0, w_done = 0, bounces = 0, q_node_active = 0, q_leaf_active = 0, q_done_active = 0, maxstack = 0;
        std::vector<v3> hitpts; std::vector<uint32_t> hittri; std::vector<float> hitthr;
        unsigned long long skips_set = 0, skips_possible = 0;
        for (uint64_t w = 0; w < nrays; w += NQ) {
            Query q[64]; Ray r[64];
            enum { NODE, LEAF, DONE, IDLE } st[64];
            int nq = (int) std::min<uint64_t>(NQ, nrays - w);
            for (int i = 0; i < 64; ++i) st[i] = IDLE;
            for (int i = 0; i < nq; ++i) {
                r[i].o = src; r[i].d = mk3(dirs[4 * (w + i)], dirs[4 * (w + i) + 1], dirs[4 * (w + i) + 2]); r[i].bounce = 0; r[i].alive = true;
                s.begin(q[i], r[i].o, r[i].d, false, 0.0f);
                st[i] = NODE;
            }
            const int hold = getenv("TRAVSIM_POSTPONE") ? atoi(getenv("TRAVSIM_POSTPONE")) : 0;   // leaves a quad may hold back (vote only)
            auto classify = [&](int i) {
                if (hold && sched == 1) {
                    // a quad that reaches a leaf holds it back and keeps walking while it may; leaves are tested when voted
                    while (q[i].ref != 0xFFFFFFFFu && (q[i].ref & RVB_BVH_LEAF) && (int) q[i].held.size() < hold && !q[i].stack.empty()) {
                        q[i].held.push_back(q[i].ref);
                        s.pop(q[i]);
                    }
                    if (q[i].ref == 0xFFFFFFFFu) st[i] = q[i].held.empty() ? DONE : LEAF;
                    else st[i] = (q[i].ref & RVB_BVH_LEAF) ? LEAF : NODE;
                    return;
                }
                if (q[i].ref == 0xFFFFFFFFu) st[i] = DONE;
                else st[i] = (q[i].ref & RVB_BVH_LEAF) ? LEAF : NODE;
            };
            int phase = 0;   // while-while emulation
            for (;;) {
                int cn = 0, cl = 0, cd = 0;
                for (int i = 0; i < nq; ++i) { cn += st[i] == NODE; cl += st[i] == LEAF; cd += st[i] == DONE; }
                if (cn + cl + cd == 0) break;
                int act;
                if (sched == 0) {
                    // node while any; then one leaf; then done; repeat
                    if (phase == 0) { if (cn) act = NODE; else { phase = 1; continue; } }
                    else if (phase == 1) { phase = 2; if (cl) act = LEAF; else continue; }
                    else { phase = 0; if (cd) act = DONE; else continue; }
                } else if (sched == 1) {
                    act = NODE; int best = cn;
                    if (cl > best) { act = LEAF; best = cl; }
                    if (cd > best) { act = DONE; best = cd; }
                } else if (sched >= 2 && sched <= 5) {
                    const double alpha = -0.25 * (sched - 1);
                    double sn = cn / pow(C_NODE, alpha), sl = cl / pow(C_LEAF, alpha), sd = cd / pow(C_DONE, alpha);
                    act = NODE; double best = sn;
                    if (sl > best) { act = LEAF; best = sl; }
                    if (sd > best) { act = DONE; best = sd; }
                } else {
                    // while-while, but the done step waits until K quads are pending (or nothing else can run)
                    const int K = sched - 5;
                    if (phase == 0) { if (cn) act = NODE; else { phase = 1; continue; } }
                    else if (phase == 1) { phase = 2; if (cl) act = LEAF; else continue; }
                    else { phase = 0; if (cd >= K || (cd && cn + cl == 0)) act = DONE; else continue; }
                }
                if (act == NODE) {
                    ++w_node; q_node_active += cn;
                    for (int i = 0; i < nq; ++i) if (st[i] == NODE) { s.node_step(q[i]); classify(i); if (q[i].stack.size() > maxstack) maxstack = q[i].stack.size(); }
                } else if (act == LEAF) {
                    ++w_leaf; q_leaf_active += cl;
                    for (int i = 0; i < nq; ++i) if (st[i] == LEAF) {
                        if (hold && sched == 1) {
                            // one leaf per step: a held one first, else the current one (then move on)
                            if (!q[i].held.empty()) { s.leaf_test(q[i], q[i].held.back()); q[i].held.pop_back(); }
                            else { s.leaf_test(q[i], q[i].ref); s.pop(q[i]); }
                            classify(i);
                            continue;
                        }
                        bool f = s.leaf_step(q[i]); if (f) st[i] = DONE; else classify(i);
                    }
                } else {
                    ++w_done; q_done_active += cd;
                    for (int i = 0; i < nq; ++i) if (st[i] == DONE) {
                        if (verify && p == 0) {
                            float bt = 0; uint32_t bi = 0xFFFFFFFFu;
                            for (const BvhTri & t : s.bs.tris) {
                                float dist = mt_intersect(mk3(t.v0[0], t.v0[1], t.v0[2]), mk3(t.e0[0], t.e0[1], t.e0[2]), mk3(t.e1[0], t.e1[1], t.e1[2]), q[i].o, q[i].d);
                                if (dist > RVB_EPSILON && (bi == 0xFFFFFFFFu || dist < bt || (dist == bt && t.index < bi))) { bt = dist; bi = t.index; }
                            }
                            ++verified;
                            if (bi != q[i].best_i || (bi != 0xFFFFFFFFu && bt != q[i].best_t)) {
                                if (++mismatches <= 10)
                                    printf("MISMATCH ray %llu bounce %u: bvh tri %u t %.9g, brute tri %u t %.9g  o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g)\n",
                                           (unsigned long long) (w + i), r[i].bounce, q[i].best_i, q[i].best_t, bi, bt, q[i].o.x, q[i].o.y, q[i].o.z, q[i].d.x, q[i].d.y, q[i].d.z);
                            }
                        }
                        if (q[i].best_i == 0xFFFFFFFFu) { st[i] = IDLE; continue; }
                        ++bounces;
                        const TriShade & sh = s.bs.shade[q[i].best_i];
                        v3 n = mk3(sh.n[0], sh.n[1], sh.n[2]);
                        v3 pnt = r[i].o + r[i].d * q[i].best_t;
                        // the product's rule (PathJob::done, trace_kernels.hip): threshold from the segment that ended here
                        const float thr = fmaf(sh.skip_b, q[i].best_t, sh.skip_a), cosine = fabsf(dot3(n, r[i].d));
                        const bool use_skip = !(getenv("TRAVSIM_SKIP") && getenv("TRAVSIM_SKIP")[0] == '0');
                        if (hitpts.size() < 400000) { hitpts.push_back(pnt); hittri.push_back(q[i].best_i); hitthr.push_back(thr); }
                        r[i].d = reflect3(n, r[i].d);
                        r[i].o = pnt;
                        if (++r[i].bounce >= nrefl) { st[i] = IDLE; continue; }
                        s.begin(q[i], r[i].o, r[i].d, false, 0.0f);
                        ++skips_possible;
                        if (use_skip && cosine > thr) { q[i].skip = sh.skip_ref; skips_set += sh.skip_ref != RVB_BVH_EMPTY; }
                        st[i] = NODE;
                    }
                }
            }
        }
        const double wn = (double) NQ * w_node / bounces, wl = (double) NQ * w_leaf / bounces, wd = (double) NQ * w_done / bounces;
        printf("sched %d T %d | ray node %.2f leaf %.2f | wave node %.2f leaf %.2f done %.2f | active node %.1f leaf %.1f done %.1f | cost/bounce %.0f\n",
               sched, T, (double) s.node_steps / bounces, (double) s.leaf_steps / bounces, wn, wl, wd,
               (double) q_node_active / w_node, (double) q_leaf_active / w_leaf, (double) q_done_active / w_done,
               wn * C_NODE + wl * C_LEAF + wd * C_DONE);
        if (p == 0) printf("   own-plane skip set on %.1f %% of the queries\n", 100.0 * skips_set / (skips_possible ? skips_possible : 1));
        if (p == 0) {
            printf("   node visits per bounce by tree level (children hit per visit):");
            for (int l = 0; l < 12; ++l) if (s.by_level[l]) printf(" L%d %.2f (%.2f)", l, (double) s.by_level[l] / bounces, (double) s.hits_by_level[l] / s.by_level[l]);
            printf("\n");
        }
        if (p > 1) continue;
        // ---- shadow_kernel replay: records grouped by leaf position, 16 consecutive records per wave pass
        {
            std::vector<uint32_t> order(hitpts.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t) i;
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return base.bs.leafpos[hittri[a]] < base.bs.leafpos[hittri[b]]; });
            const double CS_NODE = 40, CS_LEAF = 100, CS_DONE = 150;
            for (int mode = 0; mode < 3; ++mode) {       // 0 lockstep passes (shipped); here: any-hit child order 0/1/2
                Sim sh = base; sh.pol.any_order = mode;
                unsigned long long w_node = 0, w_leaf = 0, w_done = 0, an = 0, al = 0, ad = 0;
                const size_t J = 32;                      // records per quad in the job-loop modes
                const size_t per_wave = getenv("TRAVSIM_RECORDS_PER_WAVE") ? atoi(getenv("TRAVSIM_RECORDS_PER_WAVE")) : 16; (void) J;
                for (size_t w0 = 0; w0 + per_wave <= order.size(); w0 += per_wave) {
                    Query q[64];
                    enum { NODE, LEAF, DONE, IDLE } st[64];
                    size_t nextj[64];
                    auto start = [&](int i) {
                        if (nextj[i] >= 1) { st[i] = IDLE; return; }
                        const v3 pnt = hitpts[order[w0 + nextj[i] * per_wave + i]];
                        ++nextj[i];
                        v3 b2p = mic - pnt;
                        sh.begin(q[i], pnt, normalize3(b2p), true, length3(b2p));
                        {
                            const uint32_t rec = order[w0 + (nextj[i] - 1) * per_wave + i];
                            const TriShade & shd = base.bs.shade[hittri[rec]];
                            const float cosine = fabsf(dot3(mk3(shd.n[0], shd.n[1], shd.n[2]), q[i].d));       // ShadowJob::next
                            if (!(getenv("TRAVSIM_SKIP") && getenv("TRAVSIM_SKIP")[0] == '0') && cosine > hitthr[rec]) q[i].skip = shd.skip_ref;
                        }
                        st[i] = NODE;
                    };
                    for (int i = 0; i < (int) per_wave; ++i) { nextj[i] = 0; start(i); }
                    auto classify = [&](int i) {
                        if (q[i].ref == 0xFFFFFFFFu) st[i] = DONE;
                        else st[i] = (q[i].ref & RVB_BVH_LEAF) ? LEAF : NODE;
                    };
                    int phase = 0;
                    for (;;) {
                        int cn = 0, cl = 0, cd = 0;
                        for (int i = 0; i < (int) per_wave; ++i) { cn += st[i] == NODE; cl += st[i] == LEAF; cd += st[i] == DONE; }
                        if (cn + cl + cd == 0) break;
                        int act;
                        if (true) {
                            // traverse_quad per record: node while any, leaf, repeat until all finished; then one done for all
                            if (cn) act = NODE; else if (cl) act = LEAF; else act = DONE;
                        } else if (mode == 1) {
                            if (phase == 0) { if (cn) act = NODE; else { phase = 1; continue; } }
                            else if (phase == 1) { phase = 2; if (cl) act = LEAF; else continue; }
                            else { phase = 0; if (cd) act = DONE; else continue; }
                        } else {
                            act = NODE; int best = cn;
                            if (cl > best) { act = LEAF; best = cl; }
                            if (cd > best) { act = DONE; best = cd; }
                        }
                        if (act == NODE) { ++w_node; an += cn; for (int i = 0; i < (int) per_wave; ++i) if (st[i] == NODE) { sh.node_step(q[i]); classify(i); } }
                        else if (act == LEAF) { ++w_leaf; al += cl; for (int i = 0; i < (int) per_wave; ++i) if (st[i] == LEAF) { bool f = sh.leaf_step(q[i]); if (f) st[i] = DONE; else classify(i); } }
                        else {
                            ++w_done; ad += cd;
                            for (int i = 0; i < (int) per_wave; ++i) if (st[i] == DONE) {
                                if (verify && mode == 0 && p == 0) {            // any-hit against brute force (kernel.cpp:295)
                                    bool blocked = false;
                                    for (const BvhTri & t : sh.bs.tris) {
                                        float dist = mt_intersect(mk3(t.v0[0], t.v0[1], t.v0[2]), mk3(t.e0[0], t.e0[1], t.e0[2]), mk3(t.e1[0], t.e1[1], t.e1[2]), q[i].o, q[i].d);
                                        if (dist > RVB_EPSILON && dist <= q[i].tmax) { blocked = true; break; }
                                    }
                                    ++verified_any;
                                    if (blocked != q[i].found && ++mismatches_any <= 10)
                                        printf("SHADOW MISMATCH bvh %d brute %d o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) tmax %.9g\n", (int) q[i].found, (int) blocked,
                                               q[i].o.x, q[i].o.y, q[i].o.z, q[i].d.x, q[i].d.y, q[i].d.z, q[i].tmax);
                                }
                                start(i);
                            }
                        }
                    }
                }
                const double nrec = (double) (order.size() / per_wave * per_wave);
                const double wn = (double) per_wave * w_node / nrec, wl = (double) per_wave * w_leaf / nrec, wd = (double) per_wave * w_done / nrec;
                printf("   shadow mode %d | ray node %.2f leaf %.2f | wave node %.2f leaf %.2f done %.2f | active node %.1f leaf %.1f done %.1f | cost/record %.0f\n",
                       mode, sh.node_steps / nrec, sh.leaf_steps / nrec, wn, wl, wd, (double) an / w_node, (double) al / w_leaf, (double) ad / w_done,
                       wn * CS_NODE + wl * CS_LEAF + wd * CS_DONE);
            }
        }
    }
    // ---- K rays per quad (TRAVSIM_KRAYS=2,3,...): the wave holds 16 K rays; every iteration executes the step kind most QUADS can take
    // part in, and a quad takes part with whichever of its rays is in that state.  Counts wave-level steps only (no switching cost).
    if (getenv("TRAVSIM_KRAYS")) {
        for (int K = 1; K <= atoi(getenv("TRAVSIM_KRAYS")); ++K) {
            Sim s = base;
            const bool use_skip = !(getenv("TRAVSIM_SKIP") && getenv("TRAVSIM_SKIP")[0] == '0');
            unsigned long long w_node = 0, w_leaf = 0, w_done = 0, bounces = 0, a_node = 0, a_leaf = 0, a_done = 0;
            const int R = 16 * K;
            for (uint64_t w = 0; w < nrays; w += R) {
                std::vector<Query> q(R); std::vector<Ray> r(R);
                enum St { NODE, LEAF, DONE, IDLE };
                std::vector<St> st(R, IDLE);
                const int nq = (int) std::min<uint64_t>(R, nrays - w);
                for (int i = 0; i < nq; ++i) {
                    r[i].o = src; r[i].d = mk3(dirs[4 * (w + i)], dirs[4 * (w + i) + 1], dirs[4 * (w + i) + 2]); r[i].bounce = 0; r[i].alive = true;
                    s.begin(q[i], r[i].o, r[i].d, false, 0.0f);
                    st[i] = NODE;
                }
                auto classify = [&](int i) { st[i] = q[i].ref == 0xFFFFFFFFu ? DONE : ((q[i].ref & RVB_BVH_LEAF) ? LEAF : NODE); };
                for (;;) {
                    int cnt[3] = {0, 0, 0}, pick[3][16];
                    for (int quad = 0; quad < 16; ++quad)
                        for (int kind = 0; kind < 3; ++kind) {
                            pick[kind][quad] = -1;
                            for (int k = 0; k < K; ++k) { const int i = quad + 16 * k; if (i < nq && st[i] == (St) kind) { pick[kind][quad] = i; break; } }
                            cnt[kind] += pick[kind][quad] >= 0;
                        }
                    if (cnt[0] + cnt[1] + cnt[2] == 0) break;
                    int act = 0;
                    if (cnt[1] > cnt[act]) act = 1;
                    if (cnt[2] > cnt[act]) act = 2;
                    (act == 0 ? w_node : act == 1 ? w_leaf : w_done) += 1;
                    (act == 0 ? a_node : act == 1 ? a_leaf : a_done) += cnt[act];
                    for (int quad = 0; quad < 16; ++quad) {
                        const int i = pick[act][quad];
                        if (i < 0) continue;
                        if (act == 0) { s.node_step(q[i]); classify(i); }
                        else if (act == 1) { const bool f = s.leaf_step(q[i]); if (f) st[i] = DONE; else classify(i); }
                        else {
                            if (q[i].best_i == 0xFFFFFFFFu) { st[i] = IDLE; continue; }
                            ++bounces;
                            const TriShade & sh = s.bs.shade[q[i].best_i];
                            const v3 n = mk3(sh.n[0], sh.n[1], sh.n[2]);
                            const v3 pnt = r[i].o + r[i].d * q[i].best_t;
                            const float thr = fmaf(sh.skip_b, q[i].best_t, sh.skip_a), cosine = fabsf(dot3(n, r[i].d));
                            r[i].d = reflect3(n, r[i].d);
                            r[i].o = pnt;
                            if (++r[i].bounce >= nrefl) { st[i] = IDLE; continue; }
                            s.begin(q[i], r[i].o, r[i].d, false, 0.0f);
                            if (use_skip && cosine > thr) q[i].skip = sh.skip_ref;
                            st[i] = NODE;
                        }
                    }
                }
            }
            const double wn = 16.0 * w_node / bounces, wl = 16.0 * w_leaf / bounces, wd = 16.0 * w_done / bounces;
            printf("K %d rays per quad | per 16 ray-bounces: wave node %.2f leaf %.2f done %.2f | quads active node %.1f leaf %.1f done %.1f | cost %.0f\n",
                   K, wn, wl, wd, (double) a_node / w_node, (double) a_leaf / w_leaf, (double) a_done / w_done, wn * C_NODE + wl * C_LEAF + wd * C_DONE);
        }
    }
    // ---- a POOL of R rays per wave (TRAVSIM_POOL=R): any quad takes any ray; every iteration executes the step kind most rays wait for,
    // with up to 16 of them.  Counts wave-level steps only (no cost for moving ray state in and out of the quads).
    if (getenv("TRAVSIM_POOL")) {
        for (int R : {12, 13, 14, 15, 16, 24, 32, 48, 64}) {
            if (R > atoi(getenv("TRAVSIM_POOL"))) break;
            Sim s = base;
            const bool use_skip = !(getenv("TRAVSIM_SKIP") && getenv("TRAVSIM_SKIP")[0] == '0');
            unsigned long long w_step[3] = {0, 0, 0}, a_step[3] = {0, 0, 0}, bounces = 0;
            for (uint64_t w = 0; w < nrays; w += R) {
                std::vector<Query> q(R); std::vector<Ray> r(R);
                enum St { NODE, LEAF, DONE, IDLE };
                std::vector<St> st(R, IDLE);
                const int nq = (int) std::min<uint64_t>(R, nrays - w);
                for (int i = 0; i < nq; ++i) {
                    r[i].o = src; r[i].d = mk3(dirs[4 * (w + i)], dirs[4 * (w + i) + 1], dirs[4 * (w + i) + 2]); r[i].bounce = 0; r[i].alive = true;
                    s.begin(q[i], r[i].o, r[i].d, false, 0.0f);
                    st[i] = NODE;
                }
                auto classify = [&](int i) { st[i] = q[i].ref == 0xFFFFFFFFu ? DONE : ((q[i].ref & RVB_BVH_LEAF) ? LEAF : NODE); };
                for (;;) {
                    int cnt[3] = {0, 0, 0};
                    for (int i = 0; i < nq; ++i) if (st[i] != IDLE) ++cnt[st[i]];
                    if (cnt[0] + cnt[1] + cnt[2] == 0) break;
                    int act = 0;
                    if (cnt[1] > cnt[act]) act = 1;
                    if (cnt[2] > cnt[act]) act = 2;
                    int taken = 0;
                    ++w_step[act];
                    for (int i = 0; i < nq && taken < 16; ++i) {
                        if (st[i] != (St) act) continue;
                        ++taken;
                        if (act == 0) { s.node_step(q[i]); classify(i); }
                        else if (act == 1) { const bool f = s.leaf_step(q[i]); if (f) st[i] = DONE; else classify(i); }
                        else {
                            if (q[i].best_i == 0xFFFFFFFFu) { st[i] = IDLE; continue; }
                            ++bounces;
                            const TriShade & sh = s.bs.shade[q[i].best_i];
                            const v3 n = mk3(sh.n[0], sh.n[1], sh.n[2]);
                            const v3 pnt = r[i].o + r[i].d * q[i].best_t;
                            const float thr = fmaf(sh.skip_b, q[i].best_t, sh.skip_a), cosine = fabsf(dot3(n, r[i].d));
                            r[i].d = reflect3(n, r[i].d);
                            r[i].o = pnt;
                            if (++r[i].bounce >= nrefl) { st[i] = IDLE; continue; }
                            s.begin(q[i], r[i].o, r[i].d, false, 0.0f);
                            if (use_skip && cosine > thr) q[i].skip = sh.skip_ref;
                            st[i] = NODE;
                        }
                    }
                    a_step[act] += taken;
                }
            }
            const double wn = 16.0 * w_step[0] / bounces, wl = 16.0 * w_step[1] / bounces, wd = 16.0 * w_step[2] / bounces;
            printf("pool of %d rays | per 16 ray-bounces: wave node %.2f leaf %.2f done %.2f | quads active node %.1f leaf %.1f done %.1f | cost %.0f\n",
                   R, wn, wl, wd, (double) a_step[0] / w_step[0], (double) a_step[1] / w_step[1], (double) a_step[2] / w_step[2],
                   wn * C_NODE + wl * C_LEAF + wd * C_DONE);
        }
    }
    if (verify) printf("verify: %llu closest-hit queries against brute force, %llu mismatches; %llu any-hit (shadow) queries, %llu mismatches\n",
                       verified, mismatches, verified_any, mismatches_any);
    return 0;
}
