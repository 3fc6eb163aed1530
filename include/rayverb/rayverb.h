// rayverb.h — C++ host interface of the MI355X-native ray tracer: the classes and free functions
// of reference rayverb/rayverb.h (same names, argument meaning and error behaviour) implemented
// on the C-ABI of include/rvb_capi.h.  Callers written against the reference (cmd/main.cpp:241-298,
// the gtest fixtures that inherit from these classes) compile against this header instead.
//
// Differences a caller can observe are listed in INTEGRATION.md: no OpenCL objects are exposed
// (ContextProvider / KernelLoader keep their names for inheritance but hold an rvb context),
// exactly `directions.size()` rays are traced (quirk Q1), zero-volume impulses attenuate to {0, 0}
// (quirk Q2).
#pragma once

#include "clstructs.h"
#include "filters.h"
#include "generic_functions.h"

#include "rapidjson/document.h"

#include <algorithm>
#include <array>
#include <memory>
#include <numeric>
#include <string>
#include <utility>
#include <vector>

struct rvb_ctx;
struct rvb_multi;

// ---- time binning / predelay (reference rayverb.h:24-97) ----------------------------------------

// flattenImpulses: sum impulses that fall on the same sample; [8 bands][samples] (rayverb.cpp:48-77).
// Runs on the GPU; float sums are formed in impulse order, bit for bit like the reference's loop.
std::vector<std::vector<float>> flattenImpulses(const std::vector<AttenuatedImpulse> & impulse, float samplerate);
// ... mapped over channels (rayverb.cpp:28-44).
std::vector<std::vector<std::vector<float>>> flattenImpulses(const std::vector<std::vector<AttenuatedImpulse>> & impulse, float samplerate);

// Filter every band, mix the bands down per channel, then optionally normalise all channels
// together, scale, and trim the inaudible tail (reference rayverb.h:36-45, rayverb.cpp:125-149).
std::vector<std::vector<float>> process(RayverbFiltering::FilterType filtertype, std::vector<std::vector<std::vector<float>>> & data,
                                        float sr, bool do_normalize, float lo_cutoff, bool do_trim_tail, float volumme_scale);

// Earliest non-zero impulse time over any nesting of vectors (reference rayverb.h:49-74).
inline float findPredelay(const AttenuatedImpulse & i) { return i.time; }
template <typename T>
inline float findPredelay(const std::vector<T> & ret)
{
    float a = 0;
    bool first = true;
    for (const T & item : ret) {
        const float pd = findPredelay(item);
        if (first) { a = pd; first = false; }
        else if (a == 0) a = pd;
        else if (pd != 0) a = std::min(a, pd);
    }
    return a;
}

// Subtract `seconds` from every impulse time, clamping at zero (reference rayverb.h:76-90).
inline void fixPredelay(AttenuatedImpulse & ret, float seconds) { ret.time = ret.time > seconds ? ret.time - seconds : 0; }
template <typename T>
inline void fixPredelay(std::vector<T> & ret, float seconds)
{
    for (T & item : ret)
        fixPredelay(item, seconds);
}
// Find, then remove (reference rayverb.h:92-97).
template <typename T>
inline void fixPredelay(std::vector<T> & ret) { fixPredelay(ret, findPredelay(ret)); }
// The two instantiations the reference's callers use (cmd/main.cpp:292: all channels; one channel) as library functions:
// same results as the templates above, computed by several host threads, and applied as well to the device copies the
// attenuators keep of the vectors they returned, so that flattenImpulses need not upload them again.
float findPredelay(const std::vector<AttenuatedImpulse> & ret);
float findPredelay(const std::vector<std::vector<AttenuatedImpulse>> & ret);
void fixPredelay(std::vector<AttenuatedImpulse> & ret, float seconds);
void fixPredelay(std::vector<std::vector<AttenuatedImpulse>> & ret, float seconds);
void fixPredelay(std::vector<AttenuatedImpulse> & ret);
void fixPredelay(std::vector<std::vector<AttenuatedImpulse>> & ret);

// ---- device context (reference rayverb.h:99-119) ---------------------------------------------------

// Owns one GPU context.  The reference builds an OpenCL context here (rayverb.cpp:151-164); this
// one binds the rvb library to a gfx950 device (RVB_DEVICE selects which, default 0) and throws
// cl::Error when there is none — there is no CPU fallback.
class ContextProvider {
public:
    ContextProvider();
    rvb_ctx * context() const { return ctx_.get(); }
protected:
    void check(int rc, const char * where) const;     // rvb status -> cl::Error
private:
    std::shared_ptr<rvb_ctx> ctx_;
};

// The reference JIT-compiles its kernels here (rayverb.cpp:166-192); ours are ahead-of-time gfx950
// code objects inside librvb_hip.so, so this only reports the device when verbose.
class KernelLoader : public ContextProvider {
public:
    KernelLoader();
    explicit KernelLoader(bool verbose);
};

// ---- ray tracer (reference rayverb.h:121-220) -------------------------------------------------------

struct RaytracerResults {
    RaytracerResults() {}
    RaytracerResults(std::vector<Impulse> impulses, const cl_float3 & c) : impulses(std::move(impulses)), mic(c) {}
    std::vector<Impulse> impulses;
    cl_float3 mic;
};

class Raytracer : public KernelLoader {
public:
    // Own geometry (reference rayverb.h:142-148).
    Raytracer(unsigned long nreflections, std::vector<Triangle> & triangles, std::vector<cl_float3> & vertices,
              std::vector<Surface> & surfaces, bool verbose);
    // Wavefront OBJ model + JSON materials (reference rayverb.h:151-156).
    Raytracer(unsigned long nreflections, const std::string & objpath, const std::string & materialFileName, bool verbose);

    // Trace `directions.size()` rays for `nreflections` bounces (reference rayverb.h:159-164).
    void raytrace(const cl_float3 & micpos, const cl_float3 & source, const std::vector<cl_float3> & directions, bool verbose);

    RaytracerResults getRawDiffuse();                       // all nrays * nreflections slots, ray-major
    RaytracerResults getRawImages(bool removeDirect);       // de-duplicated image-source contributions
    RaytracerResults getAllRaw(bool removeDirect);          // diffuse, then images

    ~Raytracer();

private:
    struct SceneData;
    Raytracer(unsigned long nreflections, SceneData sceneData, bool verbose);
    void upload(std::vector<Triangle> & triangles, std::vector<cl_float3> & vertices, std::vector<Surface> & surfaces);
    void fetchDiffuse(std::vector<Impulse> & out);
    std::vector<Impulse> mergedImages(bool removeDirect);

    // RVB_DEVICES=N (N > 1): the trace is sharded over GPUs 0 .. N-1 of the node (rvb_multi_*, ray-range shards; results are the
    // bytes one device gives).  The reference itself drives one device (rayverb.cpp:163).
    std::shared_ptr<rvb_multi> multi_;

    const unsigned long nreflections;
    unsigned long nrays;
    std::pair<cl_float3, cl_float3> bounds;
    cl_float3 storedMicpos;
};

// ---- attenuators (reference rayverb.h:222-308) ---------------------------------------------------------

struct HrtfConfig {
    cl_float3 facing;
    cl_float3 up;
};

struct Attenuator : public KernelLoader {};

class HrtfAttenuator : public Attenuator {
public:
    HrtfAttenuator();
    // Outer vector: the two ears; inner: one AttenuatedImpulse per input impulse.
    std::vector<std::vector<AttenuatedImpulse>> attenuate(const RaytracerResults & results, const HrtfConfig & config);
    std::vector<std::vector<AttenuatedImpulse>> attenuate(const RaytracerResults & results, const cl_float3 & facing, const cl_float3 & up);

    // [channel][azimuth][elevation]; override to supply another table (reference tests/hrtf_tests.h:28).
    virtual const std::array<std::array<std::array<cl_float8, 180>, 360>, 2> & getHrtfData() const;

private:
    std::vector<AttenuatedImpulse> attenuate(const cl_float3 & mic_pos, unsigned long channel, const cl_float3 & facing,
                                             const cl_float3 & up, const std::vector<Impulse> & impulses);
};

class SpeakerAttenuator : public Attenuator {
public:
    SpeakerAttenuator();
    // Outer vector: one entry per speaker.
    std::vector<std::vector<AttenuatedImpulse>> attenuate(const RaytracerResults & results, const std::vector<Speaker> & speakers);

private:
    std::vector<AttenuatedImpulse> attenuate(const cl_float3 & mic_pos, const Speaker & speaker, const std::vector<Impulse> & impulses);
};

// Read a file and parse it as JSON into `doc`; parse errors are reported through the document
// (reference rayverb.h:311-314, rayverb.cpp:894-903).
void attemptJsonParse(const std::string & fname, rapidjson::Document & doc);
