// test_rayverb_api.cpp — the reference's three GPU gtest suites, restated against the C++ mirror
// (include/rayverb/rayverb.h) without googletest:
//   RaytracerTest.ImpulseDirections      reference tests/raytrace_tests.h:30-48, raytrace_tests.cpp:6-17
//   AttenuationTest.AttenuateSpeaker0/1/2/Timing   reference tests/attenuation_tests.h:67-101
//   HrtfTest.HrtfConfig0..3               reference tests/hrtf_tests.cpp:42-85
// Like the reference's fixtures, the test classes INHERIT from the production classes.
// TEST_OBJ / TEST_MAT are the reference's fixture files (tests/CMakeLists.txt:22-25), kept under
// tests/golden/assets.  Exit code 0 = all passed.
#include "rayverb/helpers.h"
#include "rayverb/rayverb.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>

static int failures = 0;

static bool almost_equal_4ulp(float a, float b)     // googletest's ASSERT_FLOAT_EQ
{
    if (std::isnan(a) || std::isnan(b)) return false;
    int32_t ia, ib;
    std::memcpy(&ia, &a, 4);
    std::memcpy(&ib, &b, 4);
    if (ia < 0) ia = (int32_t) 0x80000000 - ia;
    if (ib < 0) ib = (int32_t) 0x80000000 - ib;
    return std::llabs((long long) ia - ib) <= 4;
}

#define EXPECT_FLOAT_EQ_(a, b)                                                                              \
    do {                                                                                                    \
        if (!almost_equal_4ulp((a), (b))) {                                                                 \
            ++failures;                                                                                     \
            std::printf("FAIL %s:%d: %s = %.9g, expected %s = %.9g\n", __FILE__, __LINE__, #a, (double) (a), #b, (double) (b)); \
        }                                                                                                   \
    } while (0)

static void test_eq(const cl_float3 & a, const cl_float3 & b)
{
    for (int i = 0; i != 3; ++i)
        EXPECT_FLOAT_EQ_(a.s[i], b.s[i]);
}

// ---- RaytracerTest -------------------------------------------------------------------------------
class RaytracerTest : public Raytracer {
public:
    RaytracerTest() : Raytracer(NUM_REFLECTIONS, TEST_OBJ, TEST_MAT, true)
    {
        directions.push_back(cl_float3{{0, 0, -1}});
        directions.push_back(cl_float3{{0, 0, 1}});
        directions.push_back(cl_float3{{0, -1, 0}});
        directions.push_back(cl_float3{{0, 1, 0}});
        directions.push_back(cl_float3{{-1, 0, 0}});
        directions.push_back(cl_float3{{1, 0, 0}});
        directions.resize(64 * 1000, cl_float3{{0, 0, -1}});     // 16 groups of 4096 in the reference, partial last
    }
    void ImpulseDirections()
    {
        const cl_float3 mic_pos = {{0, 2, 0}}, src_pos = {{0, 2, 2}};
        raytrace(mic_pos, src_pos, directions, true);
        auto diffuse = getRawDiffuse().impulses;
        test_eq(diffuse[0 * NUM_REFLECTIONS + 0].position, cl_float3{{0, 2, -27}});
        test_eq(diffuse[1 * NUM_REFLECTIONS + 0].position, cl_float3{{0, 2, 27}});
        test_eq(diffuse[2 * NUM_REFLECTIONS + 0].position, cl_float3{{0, 0, 2}});
        test_eq(diffuse[3 * NUM_REFLECTIONS + 0].position, cl_float3{{0, 27, 2}});
        test_eq(diffuse[4 * NUM_REFLECTIONS + 0].position, cl_float3{{-25, 2, 2}});
        test_eq(diffuse[5 * NUM_REFLECTIONS + 0].position, cl_float3{{25, 2, 2}});
        test_eq(diffuse[0 * NUM_REFLECTIONS + 1].position, cl_float3{{0, 0, 0}});
        test_eq(diffuse[1 * NUM_REFLECTIONS + 1].position, cl_float3{{0, 0, 0}});
        test_eq(diffuse[2 * NUM_REFLECTIONS + 1].position, cl_float3{{0, 27, 2}});
        test_eq(diffuse[3 * NUM_REFLECTIONS + 1].position, cl_float3{{0, 0, 2}});
        test_eq(diffuse[4 * NUM_REFLECTIONS + 1].position, cl_float3{{-25, 2, -2}});
        test_eq(diffuse[5 * NUM_REFLECTIONS + 1].position, cl_float3{{25, 2, -2}});
        if (diffuse.size() != directions.size() * NUM_REFLECTIONS) { ++failures; std::printf("FAIL diffuse size\n"); }
        // API extras the reference leaves untested: images and their union with the diffuse part
        auto images = getRawImages(false).impulses;
        auto all = getAllRaw(false).impulses;
        if (images.empty() || all.size() != diffuse.size() + images.size()) { ++failures; std::printf("FAIL getAllRaw size\n"); }
        if (getRawImages(true).impulses.size() + 1 != images.size()) { ++failures; std::printf("FAIL removeDirect\n"); }
    }
    static const unsigned long NUM_REFLECTIONS = 128;
    std::vector<cl_float3> directions;
};

// ---- AttenuationTest ---------------------------------------------------------------------------------
static Impulse constructImpulse(float x, float y, float z, float time)
{
    Impulse i;
    for (int b = 0; b < 8; ++b) i.volume.s[b] = 1;
    i.position = cl_float3{{x, y, z}};
    i.time = time;
    return i;
}

static std::vector<Impulse> axisImpulses(size_t n)
{
    std::default_random_engine generator;
    std::uniform_real_distribution<float> dist(0, 100);
    std::vector<Impulse> in;
    in.push_back(constructImpulse(-10, 0, 0, dist(generator)));
    in.push_back(constructImpulse(10, 0, 0, dist(generator)));
    in.push_back(constructImpulse(0, -10, 0, dist(generator)));
    in.push_back(constructImpulse(0, 10, 0, dist(generator)));
    in.push_back(constructImpulse(0, 0, -10, dist(generator)));
    in.push_back(constructImpulse(0, 0, 10, dist(generator)));
    in.resize(n, constructImpulse(0, 0, 0, dist(generator)));
    return in;
}

class AttenuationTest : public SpeakerAttenuator {
public:
    AttenuationTest() : in(axisImpulses(1024 * 64)) {}
    void run(float shape)
    {
        const cl_float3 mic_pos = {{0, 0, 0}};
        Speaker speaker;
        speaker.direction = cl_float3{{0, 0, 1}};
        speaker.coefficient = shape;
        out = attenuate(RaytracerResults(in, mic_pos), {speaker}).front();
        for (const auto & j : out)
            for (int i = 1; i != 8; ++i)
                EXPECT_FLOAT_EQ_(j.volume.s[0], j.volume.s[i]);
    }
    void AttenuateSpeaker0() { run(0); for (const auto & j : out) EXPECT_FLOAT_EQ_(j.volume.s[0], 1.0f); }
    void AttenuateSpeaker1()
    {
        run(0.5);
        const float want[6] = {0.5f, 0.5f, 0.5f, 0.5f, 0, 1};
        for (int i = 0; i < 6; ++i) EXPECT_FLOAT_EQ_(out[i].volume.s[0], want[i]);
    }
    void AttenuateSpeaker2()
    {
        run(1);
        const float want[6] = {0, 0, 0, 0, -1, 1};
        for (int i = 0; i < 6; ++i) EXPECT_FLOAT_EQ_(out[i].volume.s[0], want[i]);
    }
    void Timing()
    {
        run(0);
        for (size_t i = 0; i < in.size(); ++i)
            if (in[i].time != out[i].time) { ++failures; std::printf("FAIL Timing at %zu\n", i); break; }
    }
    std::vector<Impulse> in;
    std::vector<AttenuatedImpulse> out;
};

// ---- HrtfTest -----------------------------------------------------------------------------------------
typedef std::array<std::array<std::array<cl_float8, 180>, 360>, 2> HrtfTable;

// the reference's test table (tests/hrtf.cpp, generated by hrtf_analysis/generate_test_hrtf_data.py):
// entry [ch][a][e] = {a, e, 0, ...} on the 15-degree grid, bilinear in between (the 360-degree
// column wraps to azimuth 0).  Only grid entries are checked.
static HrtfTable & testTable()
{
    static HrtfTable t;
    for (int ch = 0; ch < 2; ++ch)
        for (int a = 0; a < 360; ++a)
            for (int e = 0; e < 180; ++e) {
                const double a_min = 15.0 * std::floor(a / 15.0), a_max = a_min + 15.0;
                const double e_min = 15.0 * std::floor(e / 15.0), e_max = e_min + 15.0;
                const double v_lo = std::fmod(a_min, 360.0), v_hi = std::fmod(a_max, 360.0);
                cl_float8 v;
                for (int b = 0; b < 8; ++b) v.s[b] = 0;
                v.s[0] = (float) (v_lo + (v_hi - v_lo) * ((a - a_min) / 15.0));
                v.s[1] = (float) (e_min + (e_max - e_min) * ((e - e_min) / 15.0));
                t[ch][a][e] = v;
            }
    return t;
}

class HrtfTest : public HrtfAttenuator {
public:
    HrtfTest() : in(axisImpulses(1000)), HRTF_DATA(testTable()) {}
    virtual const HrtfTable & getHrtfData() const { return HRTF_DATA; }
    void run(const HrtfConfig & config)
    {
        const cl_float3 mic_pos = {{0, 0, 0}};
        out = attenuate(RaytracerResults(in, mic_pos), config).front();
    }
    void expect(int front, int back, int right, int left)
    {
        for (int i = 0; i != 8; ++i) {
            EXPECT_FLOAT_EQ_(HRTF_DATA[0][180][90].s[i], out[front].volume.s[i]);
            EXPECT_FLOAT_EQ_(HRTF_DATA[0][0][90].s[i], out[back].volume.s[i]);
            EXPECT_FLOAT_EQ_(HRTF_DATA[0][90][90].s[i], out[right].volume.s[i]);
            EXPECT_FLOAT_EQ_(HRTF_DATA[0][270][90].s[i], out[left].volume.s[i]);
        }
    }
    std::vector<Impulse> in;
    std::vector<AttenuatedImpulse> out;
    const HrtfTable & HRTF_DATA;
};

int main()
{
    try {
        { RaytracerTest t; t.ImpulseDirections(); }
        { AttenuationTest t; t.AttenuateSpeaker0(); t.AttenuateSpeaker1(); t.AttenuateSpeaker2(); t.Timing(); }
        {
            HrtfTest t;
            const cl_float3 up = {{0, 1, 0}};
            t.run(HrtfConfig{cl_float3{{0, 0, 1}}, up});  t.expect(5, 4, 0, 1);      // HrtfConfig0
            t.run(HrtfConfig{cl_float3{{1, 0, 0}}, up});  t.expect(1, 0, 5, 4);      // HrtfConfig1
            t.run(HrtfConfig{cl_float3{{0, 0, -1}}, up}); t.expect(4, 5, 1, 0);      // HrtfConfig2
            t.run(HrtfConfig{cl_float3{{-1, 0, 0}}, up}); t.expect(0, 1, 4, 5);      // HrtfConfig3
        }
        {   // flattenImpulses + predelay helpers on a tiny case with a known answer
            std::vector<AttenuatedImpulse> a(3);
            for (auto & i : a) for (int b = 0; b < 8; ++b) i.volume.s[b] = 0.25f;
            a[0].time = 0.5f; a[1].time = 0.5f; a[2].time = 0.25f;
            std::vector<std::vector<AttenuatedImpulse>> chans(1, a);
            if (findPredelay(chans) != 0.25f) { ++failures; std::printf("FAIL findPredelay\n"); }
            fixPredelay(chans);
            auto flat = flattenImpulses(chans, 8.0f);
            if (flat.size() != 1 || flat[0].size() != 8 || flat[0][0].size() != 3 || flat[0][3][0] != 0.25f || flat[0][3][2] != 0.5f)
                { ++failures; std::printf("FAIL flattenImpulses\n"); }
        }
        {   // The sequence of reference cmd/main.cpp:241-298 with the vectors handed from stage to stage (their device copies are
            // reused) must give the bytes that the same stages give on COPIES of those vectors (fresh buffers: every stage uploads).
            std::vector<cl_float3> dirs = getSeededDirections(3000, 7);
            Raytracer tracer(16, TEST_OBJ, TEST_MAT, false);
            const cl_float3 mic = {{0, 2, 0}}, src = {{0, 2, 2}};
            tracer.raytrace(mic, src, dirs, false);
            RaytracerResults results = tracer.getAllRaw(false);
            const std::vector<Speaker> speakers = {Speaker{{{-1, 0, -1, 0}}, 0.5f}, Speaker{{{1, 0, -1, 0}}, 0.5f}};
            auto same = [](const std::vector<std::vector<AttenuatedImpulse>> & x, const std::vector<std::vector<AttenuatedImpulse>> & y) {
                if (x.size() != y.size()) return false;
                for (size_t c = 0; c < x.size(); ++c) {
                    if (x[c].size() != y[c].size()) return false;
                    for (size_t i = 0; i < x[c].size(); ++i)
                        if (std::memcmp(&x[c][i].volume, &y[c][i].volume, sizeof(VolumeType)) || x[c][i].time != y[c][i].time) return false;
                }
                return true;
            };
            auto resident = SpeakerAttenuator().attenuate(results, speakers);
            RaytracerResults copy(results.impulses, results.mic);                  // same values, another buffer
            auto uploaded = SpeakerAttenuator().attenuate(copy, speakers);
            if (resident[0].size() != results.impulses.size() || !same(resident, uploaded)) { ++failures; std::printf("FAIL resident attenuate\n"); }
            auto resident_copy = resident;                                         // fresh buffers: no device copies
            fixPredelay(resident);
            fixPredelay(resident_copy);
            if (!same(resident, resident_copy)) { ++failures; std::printf("FAIL fixPredelay\n"); }
            if (flattenImpulses(resident, 44100.0f) != flattenImpulses(resident_copy, 44100.0f)) { ++failures; std::printf("FAIL resident flatten\n"); }
            // an edit of a handed-out vector must be seen (element 0 is part of the sample that guards the device copy)
            results.impulses[0].volume.s[0] = 123.0f;
            resident[1][0].volume.s[3] = -7.0f;
            RaytracerResults edited(results.impulses, results.mic);
            if (!same(SpeakerAttenuator().attenuate(results, speakers), SpeakerAttenuator().attenuate(edited, speakers)))
                { ++failures; std::printf("FAIL edited results\n"); }
            auto edited_att = resident;
            if (flattenImpulses(resident, 44100.0f) != flattenImpulses(edited_att, 44100.0f)) { ++failures; std::printf("FAIL edited attenuated\n"); }
            // The default mode uploads whatever vector a stage is handed (reference rayverb.cpp:863-875): an edit of ANY element
            // between two stages is seen.  Elements 1 and 7 are outside every sample RVB_API_RESIDENT=1 would compare (stride 251),
            // which is exactly the case that mode's contract excludes ("the caller does not edit the vectors").
            const bool promised = std::getenv("RVB_API_RESIDENT") && std::getenv("RVB_API_RESIDENT")[0] == '1';
            if (!promised) {
                tracer.raytrace(mic, src, dirs, false);
                RaytracerResults fresh = tracer.getAllRaw(false);
                const auto before = SpeakerAttenuator().attenuate(fresh, speakers);
                fresh.impulses[1].volume.s[2] = 55.0f;                             // ONE impulse, not element 0, 251, ...
                fresh.impulses[1].time = 0.125f;
                auto after = SpeakerAttenuator().attenuate(fresh, speakers);
                RaytracerResults fresh_copy(fresh.impulses, fresh.mic);
                if (!same(after, SpeakerAttenuator().attenuate(fresh_copy, speakers))) { ++failures; std::printf("FAIL one edited impulse: stale input used\n"); }
                if (same(after, before)) { ++failures; std::printf("FAIL one edited impulse: edit not seen\n"); }
                after[0][7].volume.s[5] = 3.5f;                                    // ... and one attenuated impulse before the binning
                after[0][7].time = 0.01f;
                auto after_copy = after;
                const auto flat_edit = flattenImpulses(after, 44100.0f);
                if (flat_edit != flattenImpulses(after_copy, 44100.0f)) { ++failures; std::printf("FAIL one edited attenuated impulse: stale input used\n"); }
                fixPredelay(after);
                fixPredelay(after_copy);
                after[1][7].volume.s[0] = -2.0f;
                after_copy[1][7].volume.s[0] = -2.0f;
                if (flattenImpulses(after, 44100.0f) != flattenImpulses(after_copy, 44100.0f)) { ++failures; std::printf("FAIL edit after fixPredelay: stale input used\n"); }
                // a copy of the tracer shares its context: a trace through the copy must not leave the original's results readable as current
                Raytracer twin(tracer);
                RaytracerResults mine = tracer.getRawDiffuse();
                const cl_float3 elsewhere = {{1, 2, 1}};
                twin.raytrace(elsewhere, src, dirs, false);
                RaytracerResults mine_copy(mine.impulses, mine.mic);
                if (!same(SpeakerAttenuator().attenuate(mine, speakers), SpeakerAttenuator().attenuate(mine_copy, speakers)))
                    { ++failures; std::printf("FAIL results of a tracer whose copy traced again\n"); }
            } else {
                // opt-in mode: the same copied-tracer sequence must not serve the twin's trace buffer for the original's vector
                tracer.raytrace(mic, src, dirs, false);
                Raytracer twin(tracer);
                RaytracerResults mine = tracer.getRawDiffuse();
                const cl_float3 elsewhere = {{1, 2, 1}};
                twin.raytrace(elsewhere, src, dirs, false);
                RaytracerResults mine_copy(mine.impulses, mine.mic);
                if (!same(SpeakerAttenuator().attenuate(mine, speakers), SpeakerAttenuator().attenuate(mine_copy, speakers)))
                    { ++failures; std::printf("FAIL (resident) results of a tracer whose copy traced again\n"); }
            }
            // HRTF attenuator through the same handover
            tracer.raytrace(mic, src, dirs, false);
            RaytracerResults again = tracer.getRawDiffuse();
            RaytracerResults again_copy(again.impulses, again.mic);
            const cl_float3 facing = {{0, 0, 1}}, up = {{0, 1, 0}};
            if (!same(HrtfAttenuator().attenuate(again, facing, up), HrtfAttenuator().attenuate(again_copy, facing, up)))
                { ++failures; std::printf("FAIL resident hrtf\n"); }
        }
    } catch (const cl::Error & e) {
        std::printf("cl::Error %d: %s\n", e.err(), e.what());
        return 2;
    } catch (const std::exception & e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
    std::printf(failures ? "%d FAILURES\n" : "all reference gtest cases passed (%d failures)\n", failures);
    return failures ? 1 : 0;
}
