// rayverb_api.cpp — the reference's host classes (rayverb/rayverb.h) on top of the rvb C-ABI.
// Every compute step goes through librvb_hip.so (HIP kernels); nothing here computes results on
// the CPU except the reference's own host-side steps (bounds warnings, vector plumbing).
#include "../../include/rayverb/rayverb.h"
#include "../../include/rvb_capi.h"
#include "scene_loader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>

static_assert(sizeof(Triangle) == sizeof(rvb_triangle), "Triangle");
static_assert(sizeof(cl_float3) == sizeof(rvb_float3), "cl_float3");
static_assert(sizeof(Surface) == sizeof(rvb_surface), "Surface");
static_assert(sizeof(Impulse) == sizeof(rvb_impulse), "Impulse");
static_assert(sizeof(AttenuatedImpulse) == sizeof(rvb_attenuated_impulse), "AttenuatedImpulse");
static_assert(sizeof(Speaker) == sizeof(rvb_speaker), "Speaker");

namespace {

int device_from_env()
{
    const char * e = std::getenv("RVB_DEVICE");
    return e ? std::atoi(e) : 0;
}

std::shared_ptr<rvb_ctx> make_context()
{
    rvb_ctx * raw = nullptr;
    const int rc = rvb_create(&raw, device_from_env(), 0);
    if (rc != RVB_OK)
        throw cl::Error(rc, rvb_last_error(nullptr));       // the reference throws cl::Error from cl::Context
    return std::shared_ptr<rvb_ctx>(raw, rvb_destroy);
}

// flattenImpulses is a free function in the reference; it gets a lazily created context of its own.
rvb_ctx * shared_context()
{
    static std::shared_ptr<rvb_ctx> ctx = make_context();
    return ctx.get();
}

}  // namespace

// ---- context -------------------------------------------------------------------------------------

ContextProvider::ContextProvider() : ctx_(make_context()) {}

void ContextProvider::check(int rc, const char * where) const
{
    if (rc != RVB_OK)
        throw cl::Error(rc, (std::string(where) + ": " + rvb_last_error(ctx_.get())).c_str());
}

KernelLoader::KernelLoader() : KernelLoader(false) {}

KernelLoader::KernelLoader(bool verbose)
{
    if (verbose) {
        char arch[64];
        int cus = 0;
        uint64_t hbm = 0;
        check(rvb_device_info(context(), arch, sizeof(arch), &cus, &hbm), "rvb_device_info");
        std::cerr << "rvb: " << arch << ", " << cus << " compute units, " << (hbm >> 30) << " GiB HBM (ahead-of-time kernels, no build log)"
                  << std::endl;
    }
}

// ---- time binning ----------------------------------------------------------------------------------

std::vector<std::vector<float>> flattenImpulses(const std::vector<AttenuatedImpulse> & impulse, float samplerate)
{
    rvb_ctx * ctx = shared_context();
    const rvb_attenuated_impulse * in = reinterpret_cast<const rvb_attenuated_impulse *>(impulse.data());
    uint64_t nbins = 0;
    int rc = rvb_flatten(ctx, in, impulse.size(), samplerate, nullptr, 0, &nbins);
    if (rc != RVB_OK)
        throw cl::Error(rc, rvb_last_error(ctx));
    std::vector<float> flat(8 * nbins);
    rc = rvb_flatten(ctx, in, impulse.size(), samplerate, flat.data(), nbins, &nbins);
    if (rc != RVB_OK)
        throw cl::Error(rc, rvb_last_error(ctx));
    std::vector<std::vector<float>> flattened(sizeof(VolumeType) / sizeof(float));
    for (size_t b = 0; b < flattened.size(); ++b)
        flattened[b].assign(flat.begin() + (long) (b * nbins), flat.begin() + (long) ((b + 1) * nbins));
    return flattened;
}

std::vector<std::vector<std::vector<float>>> flattenImpulses(const std::vector<std::vector<AttenuatedImpulse>> & attenuated, float samplerate)
{
    std::vector<std::vector<std::vector<float>>> flattened(attenuated.size());
    for (size_t i = 0; i < attenuated.size(); ++i)
        flattened[i] = flattenImpulses(attenuated[i], samplerate);
    return flattened;
}

// ---- ray tracer ------------------------------------------------------------------------------------

struct Raytracer::SceneData : public LoadedScene {
    SceneData(const std::string & objpath, const std::string & materialFileName, bool verbose)
        : LoadedScene(loadScene(objpath, materialFileName, verbose)) {}
};

namespace {

// reference rayverb.cpp:194-239
std::pair<cl_float3, cl_float3> getBounds(const std::vector<cl_float3> & vertices)
{
    cl_float3 lo = {{0, 0, 0, 0}}, hi = {{0, 0, 0, 0}};
    if (!vertices.empty()) {
        lo = hi = vertices.front();
        for (const cl_float3 & v : vertices)
            for (int i = 0; i < 4; ++i) {
                lo.s[i] = std::min(lo.s[i], v.s[i]);
                hi.s[i] = std::max(hi.s[i], v.s[i]);
            }
    }
    return std::make_pair(lo, hi);
}

bool inside(const std::pair<cl_float3, cl_float3> & bounds, const cl_float3 & point)
{
    // the reference loops over all four lanes of cl_float3 (rayverb.cpp:235); lane 3 is padding
    for (int i = 0; i < 3; ++i)
        if (!(bounds.first.s[i] <= point.s[i] && point.s[i] <= bounds.second.s[i]))
            return false;
    return true;
}

}  // namespace

Raytracer::Raytracer(unsigned long nreflections, std::vector<Triangle> & triangles, std::vector<cl_float3> & vertices,
                     std::vector<Surface> & surfaces, bool verbose)
    : KernelLoader(verbose), nreflections(nreflections), nrays(0), bounds(getBounds(vertices))
{
    storedMicpos = cl_float3{{0, 0, 0, 0}};
    upload(triangles, vertices, surfaces);
}

Raytracer::Raytracer(unsigned long nreflections, const std::string & objpath, const std::string & materialFileName, bool verbose)
    : Raytracer(nreflections, SceneData(objpath, materialFileName, verbose), verbose)
{
}

Raytracer::Raytracer(unsigned long nreflections, SceneData sceneData, bool verbose)
    : Raytracer(nreflections, sceneData.triangles, sceneData.vertices, sceneData.surfaces, verbose)
{
}

void Raytracer::upload(std::vector<Triangle> & triangles, std::vector<cl_float3> & vertices, std::vector<Surface> & surfaces)
{
    check(rvb_set_scene(context(), reinterpret_cast<const rvb_triangle *>(triangles.data()), triangles.size(),
                        reinterpret_cast<const rvb_float3 *>(vertices.data()), vertices.size(),
                        reinterpret_cast<const rvb_surface *>(surfaces.data()), surfaces.size()),
          "rvb_set_scene");
}

void Raytracer::raytrace(const cl_float3 & micpos, const cl_float3 & source, const std::vector<cl_float3> & directions, bool verbose)
{
    storedMicpos = micpos;

    // reference rayverb.cpp:547-583: warn when mic or source lie outside the model's bounding box
    const bool micinside = inside(bounds, micpos);
    const bool srcinside = inside(bounds, source);
    if (verbose && !(micinside && srcinside)) {
        std::cerr << "model bounds: [" << bounds.first.s[0] << ", " << bounds.first.s[1] << ", " << bounds.first.s[2] << "], ["
                  << bounds.second.s[0] << ", " << bounds.second.s[1] << ", " << bounds.second.s[2] << "]" << std::endl;
        if (!micinside) {
            std::cerr << "WARNING: microphone position may be outside model" << std::endl;
            std::cerr << "mic position: [" << micpos.s[0] << ", " << micpos.s[1] << ", " << micpos.s[2] << "]" << std::endl;
        }
        if (!srcinside) {
            std::cerr << "WARNING: source position may be outside model" << std::endl;
            std::cerr << "src position: [" << source.s[0] << ", " << source.s[1] << ", " << source.s[2] << "]" << std::endl;
        }
    }

    // air absorption per band, reference rayverb.cpp:632-641
    const float air[8] = {(float) (0.001 * -0.1), (float) (0.001 * -0.2), (float) (0.001 * -0.5), (float) (0.001 * -1.1),
                          (float) (0.001 * -2.7), (float) (0.001 * -9.4), (float) (0.001 * -29.0), (float) (0.001 * -60.0)};
    nrays = directions.size();
    check(rvb_set_directions(context(), reinterpret_cast<const rvb_float3 *>(directions.data()), directions.size()), "rvb_set_directions");
    check(rvb_trace(context(), micpos.s, source.s, nreflections, air, 0), "rvb_trace");
    check(rvb_synchronize(context()), "rvb_synchronize");     // the reference's raytrace() is blocking
}

RaytracerResults Raytracer::getRawDiffuse()
{
    std::vector<Impulse> diffuse(nrays * nreflections);
    check(rvb_get_diffuse(context(), reinterpret_cast<rvb_impulse *>(diffuse.data())), "rvb_get_diffuse");
    return RaytracerResults(diffuse, storedMicpos);
}

RaytracerResults Raytracer::getRawImages(bool removeDirect)
{
    uint64_t ncand = 0;
    check(rvb_get_image_candidates(context(), nullptr, 0, &ncand), "rvb_get_image_candidates");
    std::vector<rvb_image_candidate> cand(ncand);
    check(rvb_get_image_candidates(context(), cand.data(), cand.size(), &ncand), "rvb_get_image_candidates");
    rvb_impulse direct;
    check(rvb_get_direct(context(), &direct), "rvb_get_direct");
    const rvb_impulse * direct_ptr = nrays ? &direct : nullptr;
    uint64_t count = 0;
    check(rvb_merge_images(cand.data(), cand.size(), direct_ptr, removeDirect, nullptr, 0, &count), "rvb_merge_images");
    std::vector<Impulse> ret(count);
    check(rvb_merge_images(cand.data(), cand.size(), direct_ptr, removeDirect, reinterpret_cast<rvb_impulse *>(ret.data()), count, &count),
          "rvb_merge_images");
    return RaytracerResults(ret, storedMicpos);
}

RaytracerResults Raytracer::getAllRaw(bool removeDirect)
{
    std::vector<Impulse> diffuse = getRawDiffuse().impulses;
    const std::vector<Impulse> image = getRawImages(removeDirect).impulses;
    diffuse.insert(diffuse.end(), image.begin(), image.end());
    return RaytracerResults(diffuse, storedMicpos);
}

// ---- attenuators -------------------------------------------------------------------------------------

namespace {

typedef std::array<std::array<std::array<cl_float8, 180>, 360>, 2> HrtfTable;

// The reference's HRTF_DATA (rayverb/hrtf.cpp, derived from the IRCAM Listen database) is not part
// of the reference checkout (.MISSING_LARGE_BLOBS).  This stand-in is a smooth analytic head-shadow
// pattern with the same shape; override getHrtfData() to supply measured data.
const HrtfTable & standin_hrtf()
{
    static HrtfTable storage;                 // static storage honours the 32-byte alignment of cl_float8
    static HrtfTable * const table = &storage;
    static std::once_flag once;
    std::call_once(once, [] {
        const double pi = 3.14159265358979323846;
        for (int ch = 0; ch < 2; ++ch) {
            const double sign = ch == 0 ? -1.0 : 1.0;
            for (int a = 0; a < 360; ++a)
                for (int e = 0; e < 180; ++e) {
                    const double az = (a - 180.0) * pi / 180.0, el = (90.0 - e) * pi / 180.0;
                    const double lateral = sign * std::sin(az) * std::cos(el);
                    for (int b = 0; b < 8; ++b)
                        (*table)[ch][a][e].s[b] = (float) (0.55 + 0.45 * std::exp(-0.08 * (b + 1.0) * (1.0 - lateral)));
                }
        }
    });
    return *table;
}

}  // namespace

HrtfAttenuator::HrtfAttenuator() {}

const std::array<std::array<std::array<cl_float8, 180>, 360>, 2> & HrtfAttenuator::getHrtfData() const
{
    return standin_hrtf();
}

std::vector<std::vector<AttenuatedImpulse>> HrtfAttenuator::attenuate(const RaytracerResults & results, const HrtfConfig & config)
{
    return attenuate(results, config.facing, config.up);
}

std::vector<std::vector<AttenuatedImpulse>> HrtfAttenuator::attenuate(const RaytracerResults & results, const cl_float3 & facing, const cl_float3 & up)
{
    std::vector<std::vector<AttenuatedImpulse>> attenuated(2);          // channels {0, 1}, rayverb.cpp:751
    for (unsigned long ch = 0; ch < 2; ++ch)
        attenuated[ch] = attenuate(results.mic, ch, facing, up, results.impulses);
    return attenuated;
}

std::vector<AttenuatedImpulse> HrtfAttenuator::attenuate(const cl_float3 & mic_pos, unsigned long channel, const cl_float3 & facing,
                                                         const cl_float3 & up, const std::vector<Impulse> & impulses)
{
    // [360][180] of cl_float8 is contiguous: exactly the flattened table of rayverb.cpp:774-780
    const float * table = reinterpret_cast<const float *>(getHrtfData()[channel].data());
    std::vector<AttenuatedImpulse> ret(impulses.size());
    check(rvb_attenuate_hrtf(context(), mic_pos.s, reinterpret_cast<const rvb_impulse *>(impulses.data()), impulses.size(), table,
                             facing.s, up.s, channel, reinterpret_cast<rvb_attenuated_impulse *>(ret.data())),
          "rvb_attenuate_hrtf");
    return ret;
}

SpeakerAttenuator::SpeakerAttenuator() {}

std::vector<std::vector<AttenuatedImpulse>> SpeakerAttenuator::attenuate(const RaytracerResults & results, const std::vector<Speaker> & speakers)
{
    std::vector<std::vector<AttenuatedImpulse>> attenuated(speakers.size());
    for (size_t i = 0; i < speakers.size(); ++i)
        attenuated[i] = attenuate(results.mic, speakers[i], results.impulses);
    return attenuated;
}

std::vector<AttenuatedImpulse> SpeakerAttenuator::attenuate(const cl_float3 & mic_pos, const Speaker & speaker, const std::vector<Impulse> & impulses)
{
    std::vector<AttenuatedImpulse> ret(impulses.size());
    check(rvb_attenuate_speaker(context(), mic_pos.s, reinterpret_cast<const rvb_impulse *>(impulses.data()), impulses.size(),
                                reinterpret_cast<const rvb_speaker *>(&speaker), reinterpret_cast<rvb_attenuated_impulse *>(ret.data())),
          "rvb_attenuate_speaker");
    return ret;
}
