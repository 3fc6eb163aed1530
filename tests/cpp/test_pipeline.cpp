// test_pipeline.cpp — a C++11 caller of the impulse-response pipeline behind the C-ABI (rvb_pipeline_*, csrc/pipeline.hip): what a
// batch caller of the reference does with cmd/main.cpp:241-298 in a loop.  Twenty jobs with their own microphone and source each go
// through four contexts of the GPU; every result must equal, bit for bit, the same impulse response generated on a fifth context with
// the step-by-step calls (rvb_trace -> rvb_merge_images -> rvb_ir_configure_speakers -> rvb_ir_download, exact mode).  Then the HRTF
// model with a facing of its own per job, one context (no grouping), and the error paths.  Exit code 0 = all passed; 2 = no GPU.
//
//   test_pipeline [time <jobs>]      "time": C2-sized rays are not available here; prints ms per job of the test scene instead
#include "rvb_capi.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int failures = 0;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) { ++failures; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)
#define OK(call)                                                                      \
    do {                                                                              \
        const int rc_ = (call);                                                       \
        if (rc_ != RVB_OK) { ++failures; std::printf("FAIL %s:%d: %s -> %d\n", __FILE__, __LINE__, #call, rc_); } \
    } while (0)

// a 24 x 9 x 14 m hall whose six walls are grids of quads (two triangles each) with a few pillars: 2 700 triangles
struct Scene {
    std::vector<rvb_triangle> tris;
    std::vector<rvb_float3> verts;
    std::vector<rvb_surface> surfaces;
    void quad_grid(const float o[3], const float du[3], const float dv[3], int nu, int nv, uint64_t surface)
    {
        const uint64_t base = verts.size();
        for (int j = 0; j <= nv; ++j)
            for (int i = 0; i <= nu; ++i) {
                rvb_float3 v;
                for (int k = 0; k < 3; ++k) v.s[k] = o[k] + du[k] * i + dv[k] * j;
                v.s[3] = 0.0f;
                verts.push_back(v);
            }
        for (int j = 0; j < nv; ++j)
            for (int i = 0; i < nu; ++i) {
                const uint64_t a = base + (uint64_t) j * (nu + 1) + i, b = a + 1, c = a + nu + 1, d = c + 1;
                tris.push_back(rvb_triangle{surface, a, b, d});
                tris.push_back(rvb_triangle{surface, a, d, c});
            }
    }
    void box(const float lo[3], const float hi[3], int n, uint64_t surface)
    {
        const float sx = (hi[0] - lo[0]) / n, sy = (hi[1] - lo[1]) / n, sz = (hi[2] - lo[2]) / n;
        const float X[3] = {sx, 0, 0}, Y[3] = {0, sy, 0}, Z[3] = {0, 0, sz};
        const float p[3] = {lo[0], lo[1], lo[2]}, qx[3] = {hi[0], lo[1], lo[2]}, qy[3] = {lo[0], hi[1], lo[2]}, qz[3] = {lo[0], lo[1], hi[2]};
        quad_grid(p, X, Y, n, n, surface); quad_grid(qz, X, Y, n, n, surface);
        quad_grid(p, X, Z, n, n, surface); quad_grid(qy, X, Z, n, n, surface);
        quad_grid(p, Y, Z, n, n, surface); quad_grid(qx, Y, Z, n, n, surface);
    }
    Scene()
    {
        for (int s = 0; s < 3; ++s) {
            rvb_surface sf;
            for (int b = 0; b < 8; ++b) { sf.specular[b] = 0.97f - 0.01f * b - 0.02f * s; sf.diffuse[b] = 0.9f - 0.03f * b; }
            surfaces.push_back(sf);
        }
        const float lo[3] = {-12.0f, 0.0f, -7.0f}, hi[3] = {12.0f, 9.0f, 7.0f};
        box(lo, hi, 14, 1);
        for (int k = 0; k < 4; ++k) {
            const float cx = -7.5f + 5.0f * k;
            const float plo[3] = {cx - 0.4f, 0.0f, 2.6f}, phi[3] = {cx + 0.4f, 6.5f, 3.4f};
            box(plo, phi, 3, 2);
        }
    }
};

static std::vector<rvb_float3> directions(uint64_t n, uint64_t seed)
{
    std::vector<rvb_float3> d(n);
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double) (x >> 11) / 9007199254740992.0; };
    for (uint64_t i = 0; i < n; ++i) {
        const double z = 2.0 * next() - 1.0, th = 6.283185307179586 * next() - 3.141592653589793, r = std::sqrt(1.0 - z * z);
        d[i].s[0] = (float) (r * std::cos(th)); d[i].s[1] = (float) (r * std::sin(th)); d[i].s[2] = (float) z; d[i].s[3] = 0.0f;
    }
    return d;
}

static const float AIR[8] = {0.001f * -0.1f, 0.001f * -0.2f, 0.001f * -0.5f, 0.001f * -1.1f, 0.001f * -2.7f, 0.001f * -9.4f, 0.001f * -29.0f, 0.001f * -60.0f};

// the same impulse response by the step-by-step calls on one context
static std::vector<float> solo_ir(rvb_ctx * ctx, const float mic[3], const float src[3], uint64_t nrefl, const rvb_speaker * sp, uint64_t nsp,
                                  const float * table, const float * facing, const float * up, int mode, uint64_t * nbins_out, uint64_t * nimages_out)
{
    OK(rvb_trace(ctx, mic, src, nrefl, AIR, 0));
    uint64_t ncand = 0, nimg = 0;
    OK(rvb_get_image_candidates(ctx, nullptr, 0, &ncand));
    std::vector<rvb_image_candidate> cand(ncand);
    if (ncand) OK(rvb_get_image_candidates(ctx, cand.data(), ncand, &ncand));
    rvb_impulse direct;
    OK(rvb_get_direct(ctx, &direct));
    OK(rvb_merge_images(cand.data(), ncand, &direct, 0, nullptr, 0, &nimg));
    std::vector<rvb_impulse> images(nimg);
    if (nimg) OK(rvb_merge_images(cand.data(), ncand, &direct, 0, images.data(), nimg, &nimg));
    if (table) OK(rvb_ir_configure_hrtf(ctx, mic, table, facing, up, RVB_IR_ALL, images.data(), nimg));
    else OK(rvb_ir_configure_speakers(ctx, mic, sp, nsp, RVB_IR_ALL, images.data(), nimg));
    uint64_t nbins = 0;
    OK(rvb_ir_download(ctx, 1, 44100.0f, mode, nullptr, 0, &nbins));
    const uint64_t nch = table ? 2 : nsp;
    std::vector<float> out((size_t) (nch * 8 * nbins));
    OK(rvb_ir_download(ctx, 1, 44100.0f, mode, out.data(), nbins, &nbins));
    *nbins_out = nbins;
    *nimages_out = nimg;
    return out;
}

static void job_geometry(int i, float mic[3], float src[3], float facing[3])
{
    mic[0] = -9.0f + 0.9f * i; mic[1] = 1.5f + 0.05f * (i % 5); mic[2] = -4.0f + 0.35f * i;
    src[0] = 8.0f - 0.7f * i; src[1] = 1.7f + 0.1f * (i % 3); src[2] = -5.0f + 0.3f * ((i * 7) % 20);
    const float d[3] = {src[0] - mic[0], 0.0f, src[2] - mic[2]};
    const float l = std::sqrt(d[0] * d[0] + d[2] * d[2]);
    facing[0] = d[0] / l; facing[1] = 0.0f; facing[2] = d[2] / l;
}

int main(int argc, char ** argv)
{
    const bool timing = argc > 1 && std::strcmp(argv[1], "time") == 0;
    const int njobs = timing && argc > 2 ? std::atoi(argv[2]) : 20;
    const uint64_t nrays = timing ? 100000 : 20000, nrefl = timing ? 128 : 32;
    Scene scene;
    const std::vector<rvb_float3> dirs = directions(nrays, 3);
    const int NCTX = 4;
    rvb_ctx * ctxs[NCTX + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i <= NCTX; ++i) {
        const int rc = rvb_create(&ctxs[i], 0, 0);
        if (rc != RVB_OK) { std::printf("rvb_create: %s\n", rvb_last_error(nullptr)); return 2; }      // no GPU: there is no CPU path
        // the pipeline's contexts 1-3 read context 0's scene (rvb_share_scene); the solo context builds its own
        if (i == 0 || i == NCTX) OK(rvb_set_scene(ctxs[i], scene.tris.data(), scene.tris.size(), scene.verts.data(), scene.verts.size(), scene.surfaces.data(), scene.surfaces.size()));
        else OK(rvb_share_scene(ctxs[i], ctxs[0]));
        OK(rvb_set_directions(ctxs[i], dirs.data(), dirs.size()));
    }
    rvb_ctx * solo = ctxs[NCTX];
    rvb_speaker speakers[2];
    std::memset(speakers, 0, sizeof(speakers));
    speakers[0].direction[0] = -1.0f; speakers[0].direction[2] = -1.0f; speakers[0].coefficient = 0.5f;
    speakers[1].direction[0] = 1.0f; speakers[1].direction[2] = -1.0f; speakers[1].coefficient = 0.5f;

    // ---- speakers, exact mode, four contexts in groups of two ------------------------------------------------------------------
    rvb_pipeline * pipe = nullptr;
    OK(rvb_pipeline_create(&pipe, ctxs, NCTX, 0));
    float mic[3], src[3], facing[3];
    const float up[3] = {0.0f, 1.0f, 0.0f};
    CHECK(rvb_pipeline_submit(pipe, mic, src) == RVB_ERR_STATE);                    // not configured yet
    rvb_pipeline_result res;
    OK(rvb_pipeline_configure_speakers(pipe, speakers, 2, RVB_IR_ALL, 0, 1, 44100.0f, RVB_IR_EXACT, nrefl, AIR));
    CHECK(rvb_pipeline_next(pipe, &res) == RVB_ERR_STATE);                          // nothing pending
    const auto t0 = std::chrono::steady_clock::now();
    int submitted = 0, taken = 0;
    std::vector<std::vector<float> > got((size_t) njobs);
    std::vector<uint64_t> got_bins((size_t) njobs), got_images((size_t) njobs);
    while (taken < njobs) {
        // a caller that keeps a few jobs ahead of the results it takes
        while (submitted < njobs && rvb_pipeline_pending(pipe) < 8) {
            job_geometry(timing ? 3 : submitted, mic, src, facing);
            OK(rvb_pipeline_submit(pipe, mic, src));
            ++submitted;
        }
        OK(rvb_pipeline_next(pipe, &res));
        CHECK(res.job == (uint64_t) taken && res.nchannels == 2 && res.histogram != nullptr);
        if (!timing) got[(size_t) taken].assign(res.histogram, res.histogram + res.nchannels * 8 * res.nbins);
        got_bins[(size_t) taken] = res.nbins;
        got_images[(size_t) taken] = res.nimages;
        ++taken;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (timing) {
        std::printf("pipeline: %d jobs of %llu rays x %llu bounces, %.3f ms per job (%llu bins)\n", njobs, (unsigned long long) nrays, (unsigned long long) nrefl,
                    ms / njobs, (unsigned long long) got_bins[0]);
        rvb_pipeline_destroy(pipe);
        for (int i = 0; i <= NCTX; ++i) rvb_destroy(ctxs[i]);
        return failures ? 1 : 0;
    }
    CHECK(rvb_pipeline_pending(pipe) == 0);
    for (int i = 0; i < njobs; ++i) {
        job_geometry(i, mic, src, facing);
        uint64_t nbins = 0, nimg = 0;
        const std::vector<float> want = solo_ir(solo, mic, src, nrefl, speakers, 2, nullptr, nullptr, nullptr, RVB_IR_EXACT, &nbins, &nimg);
        CHECK(nbins == got_bins[(size_t) i] && nimg == got_images[(size_t) i]);
        CHECK(want.size() == got[(size_t) i].size() && std::memcmp(want.data(), got[(size_t) i].data(), want.size() * sizeof(float)) == 0);
        bool any = false;
        for (float v : want) any = any || v != 0.0f;
        CHECK(any);
    }
    std::printf("speakers: %d jobs through 4 contexts equal the step-by-step calls bit for bit (%.2f ms per job)\n", njobs, ms / njobs);
    rvb_pipeline_destroy(pipe);

    // ---- HRTF with a facing per job, fast mode within tolerance and exact mode bit-equal; one context (no grouping), three in groups of three
    std::vector<float> table((size_t) 2 * 360 * 180 * 8);
    for (int e = 0; e < 2; ++e)
        for (int a = 0; a < 360; ++a)
            for (int el = 0; el < 180; ++el)
                for (int b = 0; b < 8; ++b)
                    table[(((size_t) e * 360 + a) * 180 + el) * 8 + b] = 0.35f + 0.25f * std::cos(0.017453292f * (a - (e ? 90 : 270))) * std::sin(0.017453292f * el) + 0.02f * b;
    for (int shape = 0; shape < 2; ++shape) {
        const uint64_t nctx = shape ? 3 : 1, group = shape ? 3 : 0;
        OK(rvb_pipeline_create(&pipe, ctxs, nctx, group));
        OK(rvb_pipeline_configure_hrtf(pipe, table.data(), facing, up, RVB_IR_ALL, 0, 1, 44100.0f, RVB_IR_EXACT, nrefl, AIR));
        const int n = 7;
        int sent = 0;
        for (int i = 0; i < n; ++i) {
            while (sent < n && rvb_pipeline_pending(pipe) < 4 * nctx) {     // (as many as the pipeline takes: 4 x contexts)
                job_geometry(sent + 2, mic, src, facing);
                OK(rvb_pipeline_submit_oriented(pipe, mic, src, facing, up));
                ++sent;
            }
            OK(rvb_pipeline_next(pipe, &res));
            job_geometry(i + 2, mic, src, facing);
            uint64_t nbins = 0, nimg = 0;
            const std::vector<float> want = solo_ir(solo, mic, src, nrefl, nullptr, 0, table.data(), facing, up, RVB_IR_EXACT, &nbins, &nimg);
            CHECK(res.job == (uint64_t) i && res.nbins == nbins && res.nimages == nimg && res.nchannels == 2);
            CHECK(std::memcmp(want.data(), res.histogram, want.size() * sizeof(float)) == 0);
        }
        std::printf("hrtf: %d jobs through %llu context(s) equal the step-by-step calls bit for bit\n", n, (unsigned long long) nctx);
        rvb_pipeline_destroy(pipe);
    }

    // ---- error paths ----------------------------------------------------------------------------------------------------------------
    rvb_ctx * twice[2] = {ctxs[0], ctxs[0]};
    CHECK(rvb_pipeline_create(&pipe, twice, 2, 0) == RVB_ERR_INVALID);
    CHECK(rvb_pipeline_create(&pipe, ctxs, 0, 0) == RVB_ERR_INVALID);
    OK(rvb_pipeline_create(&pipe, ctxs, 2, 0));
    CHECK(rvb_pipeline_configure_speakers(pipe, speakers, 0, RVB_IR_ALL, 0, 1, 44100.0f, RVB_IR_EXACT, nrefl, AIR) == RVB_ERR_INVALID);
    OK(rvb_pipeline_configure_speakers(pipe, speakers, 2, RVB_IR_DIFFUSE, 0, 0, 44100.0f, RVB_IR_FAST, nrefl, AIR));
    job_geometry(1, mic, src, facing);
    for (int i = 0; i < 8; ++i) OK(rvb_pipeline_submit(pipe, mic, src));
    CHECK(rvb_pipeline_submit(pipe, mic, src) == RVB_ERR_CAPACITY);                 // 4 x contexts pending
    CHECK(rvb_pipeline_configure_speakers(pipe, speakers, 2, RVB_IR_ALL, 0, 1, 44100.0f, RVB_IR_EXACT, nrefl, AIR) == RVB_ERR_STATE);
    std::vector<float> first;
    for (int i = 0; i < 8; ++i) {
        OK(rvb_pipeline_next(pipe, &res));
        CHECK(res.predelay == 0.0f && res.nimages == 0);                            // diffuse only, no predelay trimming
        if (i == 0) first.assign(res.histogram, res.histogram + 16 * res.nbins);
        else {
            // float atomics: the same job again within the fast mode's tolerance of the first
            double worst = 0.0, peak = 0.0;
            for (size_t k = 0; k < first.size(); ++k) { worst = std::fmax(worst, std::fabs((double) first[k] - res.histogram[k])); peak = std::fmax(peak, std::fabs((double) first[k])); }
            CHECK(first.size() == 16 * res.nbins && worst <= 1e-5 * peak && peak > 0.0);
        }
    }
    rvb_pipeline_destroy(pipe);
    for (int i = 0; i <= NCTX; ++i) rvb_destroy(ctxs[i]);
    if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
    std::printf("all pipeline checks passed\n");
    return 0;
}
