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
#include <list>
#include <map>
#include <mutex>
#include <thread>

#include <sys/mman.h>

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

int devices_from_env()
{
    const char * e = std::getenv("RVB_DEVICES");
    return e ? std::max(1, std::atoi(e)) : 1;
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

// A std::vector of n elements whose storage is NOT value-initialised: the vectors this API returns hold up to a gigabyte that
// the next line overwrites from the device, and std::vector<T>(n) would first zero-fill it on one core (0.1-0.15 s per
// 819 MB vector).  libstdc++ only (the size is set through the vector base's own pointers); any other standard library gets
// the ordinary zero-filled vector.
template <class T>
std::vector<T> uninitialized_vector(size_t n)
{
#if defined(__GLIBCXX__) && !defined(_GLIBCXX_DEBUG) && !defined(RVB_SAFE_VECTORS)
    struct Access : std::vector<T> {
        void adopt(size_t count) { this->_M_impl._M_finish = this->_M_impl._M_start + count; }
    };
    static_assert(sizeof(Access) == sizeof(std::vector<T>), "no state of its own");
    std::vector<T> v;
    v.reserve(n);
    static_cast<Access &>(v).adopt(n);         // T is a POD of this library (Impulse, AttenuatedImpulse)
    // the buffer is fresh address space that the download is about to touch for the first time: ask for 2 MiB pages (400 page
    // faults instead of 200 000 per 819 MB where the kernel grants them)
    static const bool huge = !(std::getenv("RVB_HUGE_PAGES") && std::getenv("RVB_HUGE_PAGES")[0] == '0');
    if (huge && n * sizeof(T) >= (8u << 20)) {
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(v.data()) + (2u << 20) - 1) & ~(uintptr_t) ((2u << 20) - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(v.data()) + n * sizeof(T)) & ~(uintptr_t) ((2u << 20) - 1);
        if (hi > lo) (void) madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    return v;
#else
    return std::vector<T>(n);
#endif
}

unsigned host_threads()
{
    static const unsigned n = [] {
        if (const char * e = std::getenv("RVB_HOST_THREADS")) return (unsigned) std::max(1, std::atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return std::max(1u, std::min(8u, hw ? hw / 2 : 4u));
    }();
    return n;
}

// f(first, last) over [0, n) on several threads
template <class F>
void parallel_ranges(size_t n, F f)
{
    const size_t threads = n < (1u << 16) ? 1 : host_threads();
    if (threads <= 1) { f((size_t) 0, n); return; }
    std::vector<std::thread> pool;
    const size_t per = (n + threads - 1) / threads;
    for (size_t t = 0; t < threads; ++t) {
        const size_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        pool.emplace_back([=] { f(lo, hi); });
    }
    for (std::thread & t : pool) t.join();
}

// ---- device copies of vectors this library handed out (OPT-IN: RVB_API_RESIDENT=1) ---------------------------------------
// The reference's API moves every stage's result through std::vector and its implementation uploads whatever vector it is handed,
// stage by stage (rayverb.cpp:863-875).  That is what this mirror does by default: every stage uploads the vector it receives.
//
// With RVB_API_RESIDENT=1 in the environment the CALLER PROMISES not to edit a vector this library returned between the
// stages of the sequence Raytracer::getAllRaw -> Attenuator::attenuate -> fixPredelay -> flattenImpulses (cmd/main.cpp:241-298
// does not).  The producer of a vector then remembers where its device copy lives, keyed by the vector's buffer (address,
// element count), and the next stage uses that copy instead of uploading.  A changed length or address is caught by the key and a
// sample of the contents (every kSampleStride-th element) is compared as a courtesy, but an in-place edit that misses the sample
// is NOT seen — checking all of a 0.8 GB vector costs as much host-memory traffic as uploading it, which is why the mechanism
// is a promise the caller opts into and not the default.
const size_t kSampleStride = 251;
const size_t kResidentByteCap = (size_t) 6 << 30;      // owned device copies kept at a time (bytes), oldest dropped first

struct Resident {
    const void * host = nullptr;      // the vector's buffer
    size_t count = 0, elem = 0;       // elements, bytes per element
    void * device = nullptr;          // device copy (of `device_count` leading elements)
    size_t device_count = 0;
    bool owned = false;               // allocated for this entry (freed with it) / borrowed from a tracing context
    const rvb_ctx * owner = nullptr;  // borrowed: the context whose trace buffer this is (copies of a Raytracer share it) ...
    uint64_t generation = 0;          // ... and that context's trace count when the entry was made: a later trace voids the entry
    std::vector<unsigned char> sample;
};

std::mutex g_resident_mutex;
std::list<Resident> g_resident;
std::map<const rvb_ctx *, uint64_t> g_trace_generation;       // traces started per context (guarded by g_resident_mutex)

bool resident_enabled()
{
    static const bool on = std::getenv("RVB_API_RESIDENT") && std::getenv("RVB_API_RESIDENT")[0] == '1';
    return on;
}

std::vector<unsigned char> take_sample(const void * host, size_t count, size_t elem)
{
    std::vector<unsigned char> s;
    s.reserve((count / kSampleStride + 2) * elem);
    const unsigned char * p = static_cast<const unsigned char *>(host);
    for (size_t i = 0; i < count; i += kSampleStride) s.insert(s.end(), p + i * elem, p + (i + 1) * elem);
    if (count) s.insert(s.end(), p + (count - 1) * elem, p + count * elem);
    return s;
}

void release_entry(Resident & r)
{
    if (r.owned && r.device) (void) rvb_device_free(shared_context(), r.device);
    r.device = nullptr;
}

// a context is about to trace (or goes away): its trace buffer no longer holds what the borrowed entries describe
void resident_new_trace(const rvb_ctx * owner)
{
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    ++g_trace_generation[owner];
    for (auto it = g_resident.begin(); it != g_resident.end();)
        if (!it->owned && it->owner == owner) it = g_resident.erase(it); else ++it;
}

uint64_t resident_generation(const rvb_ctx * owner)
{
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    return g_trace_generation[owner];
}

// drops owned device copies, oldest first, until at most `keep_bytes` of them remain; returns whether anything was freed
bool resident_evict(size_t keep_bytes)
{
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    size_t held = 0;
    for (const Resident & r : g_resident) if (r.owned) held += r.device_count * r.elem;
    bool freed = false;
    while (held > keep_bytes && !g_resident.empty()) {
        auto it = g_resident.end();
        for (auto j = g_resident.begin(); j != g_resident.end(); ++j) if (j->owned) it = j;      // the last owned entry = the oldest
        if (it == g_resident.end()) break;
        held -= it->device_count * it->elem;
        release_entry(*it);
        g_resident.erase(it);
        freed = true;
    }
    return freed;
}

void resident_remember(Resident r)
{
    if (!resident_enabled()) { release_entry(r); return; }
    r.sample = take_sample(r.host, r.count, r.elem);
    {
        std::lock_guard<std::mutex> lock(g_resident_mutex);
        for (auto it = g_resident.begin(); it != g_resident.end();)      // a buffer address names one vector at a time
            if (it->host == r.host) { release_entry(*it); it = g_resident.erase(it); } else ++it;
        g_resident.push_front(std::move(r));
    }
    (void) resident_evict(kResidentByteCap);
}

// the entry of (host, count) if the vector still reads as it did; stale entries are dropped
bool resident_find(const void * host, size_t count, size_t elem, Resident & out)
{
    if (!resident_enabled() || !host) return false;
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    for (auto it = g_resident.begin(); it != g_resident.end(); ++it) {
        if (it->host != host) continue;
        const bool current = it->owned || g_trace_generation[it->owner] == it->generation;
        if (current && it->count == count && it->elem == elem && it->sample == take_sample(host, count, elem)) { out = *it; out.sample.clear(); return true; }
        release_entry(*it);
        g_resident.erase(it);
        return false;
    }
    return false;
}

void resident_resample(const void * host, size_t count, size_t elem)
{
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    for (Resident & r : g_resident)
        if (r.host == host && r.count == count && r.elem == elem) r.sample = take_sample(host, count, elem);
}

// rvb_device_alloc that gives the resident cache's device copies back before it reports a failure
int device_alloc_evicting(rvb_ctx * ctx, uint64_t bytes, void ** d_ptr)
{
    int rc = rvb_device_alloc(ctx, bytes, d_ptr);
    if (rc != RVB_OK && resident_evict(0))
        rc = rvb_device_alloc(ctx, bytes, d_ptr);
    return rc;
}

void throw_on(int rc, rvb_ctx * ctx, const char * where)
{
    if (rc != RVB_OK)
        throw cl::Error(rc, (std::string(where) + ": " + rvb_last_error(ctx)).c_str());
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

// ---- time binning and predelay -------------------------------------------------------------------------

std::vector<std::vector<float>> flattenImpulses(const std::vector<AttenuatedImpulse> & impulse, float samplerate)
{
    rvb_ctx * ctx = shared_context();
    uint64_t nbins = 0;
    std::vector<float> flat;
    Resident r;
    if (resident_find(impulse.data(), impulse.size(), sizeof(AttenuatedImpulse), r) && r.device_count == impulse.size()) {
        // the attenuator's device copy (kept in step by fixPredelay): no upload
        throw_on(rvb_flatten_device(ctx, r.device, impulse.size(), samplerate, nullptr, 0, &nbins), ctx, "rvb_flatten_device");
        flat.resize(8 * nbins);
        // (the size query left the keys on the device; the second call redoes only them, not an upload)
        throw_on(rvb_flatten_device(ctx, r.device, impulse.size(), samplerate, flat.data(), nbins, &nbins), ctx, "rvb_flatten_device");
    } else {
        const rvb_attenuated_impulse * in = reinterpret_cast<const rvb_attenuated_impulse *>(impulse.data());
        throw_on(rvb_flatten(ctx, in, impulse.size(), samplerate, nullptr, 0, &nbins), ctx, "rvb_flatten");      // uploads once ...
        flat.resize(8 * nbins);
        throw_on(rvb_flatten(ctx, in, impulse.size(), samplerate, flat.data(), nbins, &nbins), ctx, "rvb_flatten");   // ... and fills from the device copy
    }
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

// reference rayverb.h:49-74: the smallest non-zero time, 0 if there is none (min is order-independent: threads may split the range)
float findPredelay(const std::vector<AttenuatedImpulse> & ret)
{
    const size_t n = ret.size();
    const size_t threads = n < (1u << 16) ? 1 : host_threads();
    std::vector<float> part(threads, 0.0f);
    const size_t per = threads ? (n + threads - 1) / threads : 0;
    std::vector<std::thread> pool;
    auto work = [&](size_t t) {
        float a = 0.0f;
        for (size_t i = t * per; i < std::min(n, (t + 1) * per); ++i) {
            const float pd = ret[i].time;
            if (pd != 0.0f && (a == 0.0f || pd < a)) a = pd;
        }
        part[t] = a;
    };
    if (threads <= 1) { if (threads) work(0); }
    else {
        for (size_t t = 0; t < threads; ++t) pool.emplace_back(work, t);
        for (std::thread & t : pool) t.join();
    }
    float a = 0.0f;
    for (float pd : part)
        if (pd != 0.0f && (a == 0.0f || pd < a)) a = pd;
    return a;
}

float findPredelay(const std::vector<std::vector<AttenuatedImpulse>> & ret)
{
    float a = 0.0f;
    for (const std::vector<AttenuatedImpulse> & channel : ret) {
        const float pd = findPredelay(channel);
        if (pd != 0.0f && (a == 0.0f || pd < a)) a = pd;
    }
    return a;
}

// reference rayverb.h:76-90, on the host vector and on its device copy if there is one
void fixPredelay(std::vector<AttenuatedImpulse> & ret, float seconds)
{
    // the device copy takes part only if the vector still reads as it did when the copy was made (sample comparison)
    Resident r;
    const bool have = resident_find(ret.data(), ret.size(), sizeof(AttenuatedImpulse), r) && r.owned && r.device_count == ret.size();
    AttenuatedImpulse * a = ret.data();
    parallel_ranges(ret.size(), [a, seconds](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) a[i].time = a[i].time > seconds ? a[i].time - seconds : 0;
    });
    if (!have) return;
    rvb_ctx * ctx = shared_context();
    throw_on(rvb_fix_predelay_device(ctx, r.device, r.device_count, seconds), ctx, "rvb_fix_predelay_device");
    resident_resample(ret.data(), ret.size(), sizeof(AttenuatedImpulse));
}

void fixPredelay(std::vector<std::vector<AttenuatedImpulse>> & ret, float seconds)
{
    for (std::vector<AttenuatedImpulse> & channel : ret)
        fixPredelay(channel, seconds);
}

void fixPredelay(std::vector<AttenuatedImpulse> & ret) { fixPredelay(ret, findPredelay(ret)); }
void fixPredelay(std::vector<std::vector<AttenuatedImpulse>> & ret) { fixPredelay(ret, findPredelay(ret)); }

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
    const int ndev = devices_from_env();
    if (ndev > 1) {
        rvb_multi * raw = nullptr;
        const int rc = rvb_multi_create(&raw, nullptr, ndev, 0);
        if (rc != RVB_OK)
            throw cl::Error(rc, rvb_last_error(nullptr));
        multi_ = std::shared_ptr<rvb_multi>(raw, rvb_multi_destroy);
        const int rc2 = rvb_multi_set_scene(raw, reinterpret_cast<const rvb_triangle *>(triangles.data()), triangles.size(),
                                            reinterpret_cast<const rvb_float3 *>(vertices.data()), vertices.size(),
                                            reinterpret_cast<const rvb_surface *>(surfaces.data()), surfaces.size());
        if (rc2 != RVB_OK)
            throw cl::Error(rc2, (std::string("rvb_multi_set_scene: ") + rvb_multi_last_error(raw)).c_str());
        return;
    }
    check(rvb_set_scene(context(), reinterpret_cast<const rvb_triangle *>(triangles.data()), triangles.size(),
                        reinterpret_cast<const rvb_float3 *>(vertices.data()), vertices.size(),
                        reinterpret_cast<const rvb_surface *>(surfaces.data()), surfaces.size()),
          "rvb_set_scene");
}

Raytracer::~Raytracer()
{
    // entries that borrow the context's trace buffer: void once the context may go away (copies of this object share the
    // context; treating every destruction as the end of the buffer only costs an upload)
    resident_new_trace(context());
}

void Raytracer::raytrace(const cl_float3 & micpos, const cl_float3 & source, const std::vector<cl_float3> & directions, bool verbose)
{
    storedMicpos = micpos;
    resident_new_trace(context());             // the trace buffer is about to be overwritten (copies of this object share it)

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
    if (multi_) {
        int rc = rvb_multi_set_directions(multi_.get(), reinterpret_cast<const rvb_float3 *>(directions.data()), directions.size());
        if (rc == RVB_OK) rc = rvb_multi_trace(multi_.get(), micpos.s, source.s, nreflections, air);     // blocking, every device at once
        if (rc != RVB_OK)
            throw cl::Error(rc, (std::string("rvb_multi_trace: ") + rvb_multi_last_error(multi_.get())).c_str());
        return;
    }
    check(rvb_set_directions(context(), reinterpret_cast<const rvb_float3 *>(directions.data()), directions.size()), "rvb_set_directions");
    check(rvb_trace(context(), micpos.s, source.s, nreflections, air, 0), "rvb_trace");
    check(rvb_synchronize(context()), "rvb_synchronize");     // the reference's raytrace() is blocking
}

// diffuse impulses into out[0 .. nrays * nreflections), and the note that this buffer has a device copy
void Raytracer::fetchDiffuse(std::vector<Impulse> & out)
{
    const size_t n = nrays * nreflections;
    if (multi_) {                              // every device writes its slice of the array; no single device copy to remember
        const int rc = rvb_multi_get_diffuse(multi_.get(), reinterpret_cast<rvb_impulse *>(out.data()));
        if (rc != RVB_OK)
            throw cl::Error(rc, (std::string("rvb_multi_get_diffuse: ") + rvb_multi_last_error(multi_.get())).c_str());
        return;
    }
    check(rvb_get_diffuse(context(), reinterpret_cast<rvb_impulse *>(out.data())), "rvb_get_diffuse");
    const void * d = nullptr;
    uint64_t count = 0;
    if (n && rvb_diffuse_device(context(), &d, &count) == RVB_OK && count == n) {
        Resident r;
        r.host = out.data(); r.count = out.size(); r.elem = sizeof(Impulse);
        r.device = const_cast<void *>(d); r.device_count = n; r.owned = false; r.owner = context();
        r.generation = resident_generation(context());
        resident_remember(r);
    }
}

std::vector<Impulse> Raytracer::mergedImages(bool removeDirect)
{
    if (multi_) {
        uint64_t count = 0;
        int rc = rvb_multi_get_images(multi_.get(), removeDirect, nullptr, 0, &count);
        std::vector<Impulse> ret(count);
        if (rc == RVB_OK) rc = rvb_multi_get_images(multi_.get(), removeDirect, reinterpret_cast<rvb_impulse *>(ret.data()), count, &count);
        if (rc != RVB_OK)
            throw cl::Error(rc, (std::string("rvb_multi_get_images: ") + rvb_multi_last_error(multi_.get())).c_str());
        return ret;
    }
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
    return ret;
}

RaytracerResults Raytracer::getRawDiffuse()
{
    std::vector<Impulse> diffuse = uninitialized_vector<Impulse>(nrays * nreflections);
    fetchDiffuse(diffuse);
    return RaytracerResults(std::move(diffuse), storedMicpos);
}

RaytracerResults Raytracer::getRawImages(bool removeDirect)
{
    return RaytracerResults(mergedImages(removeDirect), storedMicpos);
}

RaytracerResults Raytracer::getAllRaw(bool removeDirect)
{
    // diffuse, then images (reference rayverb.cpp:708-714) — built in one buffer, no intermediate vectors
    const std::vector<Impulse> image = mergedImages(removeDirect);
    const size_t nd = nrays * nreflections;
    std::vector<Impulse> all = uninitialized_vector<Impulse>(nd + image.size());
    std::copy(image.begin(), image.end(), all.begin() + (long) nd);
    fetchDiffuse(all);
    return RaytracerResults(std::move(all), storedMicpos);
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

// The impulses of `results` on the shared context's device: the trace buffer they were downloaded from when the vector is
// the one getAllRaw / getRawDiffuse returned (plus the few image impulses behind it), else an upload.
namespace {
struct DeviceImpulses {
    explicit DeviceImpulses(rvb_ctx * c) : ctx(c) {}
    DeviceImpulses(const DeviceImpulses &) = delete;
    DeviceImpulses & operator=(const DeviceImpulses &) = delete;
    rvb_ctx * ctx;
    const void * head = nullptr;      // first `nhead` impulses (a Raytracer's trace buffer, borrowed, or the whole upload)
    size_t nhead = 0;
    void * tail = nullptr;            // the rest (uploaded; owned)
    size_t ntail = 0;
    void * upload = nullptr;          // owned storage behind `head` when nothing was resident
    ~DeviceImpulses()
    {
        if (tail) (void) rvb_device_free(ctx, tail);
        if (upload) (void) rvb_device_free(ctx, upload);
    }
};

void stage_impulses(const std::vector<Impulse> & impulses, DeviceImpulses & dev)
{
    rvb_ctx * ctx = dev.ctx;
    const size_t n = impulses.size();
    Resident r;
    if (resident_find(impulses.data(), n, sizeof(Impulse), r) && !r.owned && r.device_count <= n) {
        // getRawDiffuse / getAllRaw: the diffuse impulses are still in the tracer's buffer; the few image impulses that
        // getAllRaw put behind them come from the host vector
        dev.head = r.device;
        dev.nhead = r.device_count;
        dev.ntail = n - r.device_count;
        if (dev.ntail) {
            throw_on(device_alloc_evicting(ctx, dev.ntail * sizeof(Impulse), &dev.tail), ctx, "rvb_device_alloc");
            throw_on(rvb_copy_to_device(ctx, dev.tail, impulses.data() + dev.nhead, dev.ntail * sizeof(Impulse)), ctx, "rvb_copy_to_device");
        }
        return;
    }
    if (n == 0) return;
    throw_on(device_alloc_evicting(ctx, n * sizeof(Impulse), &dev.upload), ctx, "rvb_device_alloc");
    throw_on(rvb_copy_to_device(ctx, dev.upload, impulses.data(), n * sizeof(Impulse)), ctx, "rvb_copy_to_device");
    dev.head = dev.upload;
    dev.nhead = n;
}

// one channel: kernel on the staged input, result downloaded into a fresh vector whose device copy is remembered
template <class Launch>
std::vector<AttenuatedImpulse> attenuate_channel(rvb_ctx * ctx, size_t n, Launch launch)
{
    std::vector<AttenuatedImpulse> ret = uninitialized_vector<AttenuatedImpulse>(n);
    if (n == 0) return ret;
    void * d_out = nullptr;
    throw_on(device_alloc_evicting(ctx, n * sizeof(AttenuatedImpulse), &d_out), ctx, "rvb_device_alloc");
    int rc = launch(d_out);
    if (rc == RVB_OK) rc = rvb_copy_to_host(ctx, ret.data(), d_out, n * sizeof(AttenuatedImpulse));
    if (rc != RVB_OK) { (void) rvb_device_free(ctx, d_out); throw_on(rc, ctx, "attenuate"); }
    Resident r;
    r.host = ret.data(); r.count = n; r.elem = sizeof(AttenuatedImpulse);
    r.device = d_out; r.device_count = n; r.owned = true;
    resident_remember(r);                       // (frees d_out itself when residency is off)
    return ret;
}
}  // namespace

std::vector<std::vector<AttenuatedImpulse>> HrtfAttenuator::attenuate(const RaytracerResults & results, const cl_float3 & facing, const cl_float3 & up)
{
    // channels {0, 1}, rayverb.cpp:751.  The kernels run on the library's shared context: this object is usually a temporary
    // (cmd/main.cpp:286) and the device copies of what it returns have to outlive it.
    DeviceImpulses dev(shared_context());
    stage_impulses(results.impulses, dev);
    std::vector<std::vector<AttenuatedImpulse>> attenuated(2);
    for (unsigned long ch = 0; ch < 2; ++ch) {
        // [360][180] of cl_float8 is contiguous: exactly the flattened table of rayverb.cpp:774-780
        const float * table = reinterpret_cast<const float *>(getHrtfData()[ch].data());
        const cl_float3 mic = results.mic;
        const size_t n = results.impulses.size();
        attenuated[ch] = attenuate_channel(dev.ctx, n, [&](void * d_out) {
            int rc = rvb_attenuate_hrtf_device(dev.ctx, mic.s, dev.head, dev.nhead, table, facing.s, up.s, ch, d_out);
            if (rc == RVB_OK && dev.ntail)
                rc = rvb_attenuate_hrtf_device(dev.ctx, mic.s, dev.tail, dev.ntail, table, facing.s, up.s, ch,
                                               static_cast<char *>(d_out) + dev.nhead * sizeof(AttenuatedImpulse));
            return rc;
        });
    }
    return attenuated;
}

std::vector<AttenuatedImpulse> HrtfAttenuator::attenuate(const cl_float3 & mic_pos, unsigned long channel, const cl_float3 & facing,
                                                         const cl_float3 & up, const std::vector<Impulse> & impulses)
{
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
    // the impulse array is staged ONCE for all speakers (the reference uploads it per speaker, rayverb.cpp:863-875)
    DeviceImpulses dev(shared_context());
    stage_impulses(results.impulses, dev);
    std::vector<std::vector<AttenuatedImpulse>> attenuated(speakers.size());
    const size_t n = results.impulses.size();
    for (size_t i = 0; i < speakers.size(); ++i) {
        const cl_float3 mic = results.mic;
        const rvb_speaker * sp = reinterpret_cast<const rvb_speaker *>(&speakers[i]);
        attenuated[i] = attenuate_channel(dev.ctx, n, [&](void * d_out) {
            int rc = rvb_attenuate_speaker_device(dev.ctx, mic.s, dev.head, dev.nhead, sp, d_out);
            if (rc == RVB_OK && dev.ntail)
                rc = rvb_attenuate_speaker_device(dev.ctx, mic.s, dev.tail, dev.ntail, sp, static_cast<char *>(d_out) + dev.nhead * sizeof(AttenuatedImpulse));
            return rc;
        });
    }
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
