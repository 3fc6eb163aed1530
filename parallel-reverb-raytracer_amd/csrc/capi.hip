// capi.hip — implementation of include/rvb_capi.h: context, HBM buffers, call order.
// No compute happens here and nothing falls back to the CPU: every entry point that produces
// results launches the HIP kernels of trace_kernels.hip / stream_kernels.hip.
#include "../../include/rvb_capi.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void * p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
    template <class T> T * as() const { return reinterpret_cast<T *>(p); }
};

struct Timing { std::string name; hipEvent_t start, stop; };

}  // namespace

// The scene's device buffers: built and uploaded once (rvb_set_scene), read by every context that holds the store (rvb_share_scene) —
// one copy in HBM, and ONE copy for the L2s to keep, however many contexts trace in it side by side.  Freed with its last holder.
struct SceneStore {
    int device = 0;
    DevBuf nodes, tris, shade, corners, surfaces;
    SceneStore() = default;
    SceneStore(const SceneStore &) = delete;
    SceneStore & operator=(const SceneStore &) = delete;
    ~SceneStore()
    {
        (void) hipSetDevice(device);
        for (DevBuf * b : {&nodes, &tris, &shade, &corners, &surfaces})
            b->release();
    }
};

struct rvb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;          // image_kernel runs here, beside the record grouping
    hipStream_t export_stream = nullptr;        // rvb_copy_to_pinned_host_async: results leave for the host beside the next trace
    hipEvent_t export_ready = nullptr;
    hipEvent_t path_done = nullptr, side_done = nullptr;
    hipEvent_t prep_done = nullptr, group_done = nullptr;      // rvb_trace_group: this context's fills are enqueued / the group's path kernel is
    std::string error;
    std::string arch;
    int compute_units = 0;
    uint64_t hbm_bytes = 0;

    // scene
    bool have_scene = false;
    std::shared_ptr<SceneStore> store;          // (its own after rvb_set_scene, another context's after rvb_share_scene)
    SceneDev scene;
    uint64_t nnodes = 0, kept = 0;
    uint32_t depth = 0;
    uint32_t stack_need = RVB_BVH_STACK;
    uint64_t nsurfaces = 0;

    // rays + trace results
    DevBuf directions_own;
    const float4 * directions = nullptr;
    uint64_t nrays = 0;
    uint32_t concurrent_traces = 1;             // rvb_set_concurrent_traces
    uint32_t path_lanes = 0;                    // rvb_set_path_lanes: 0 = chosen per launch (rvb_path_lanes_for)
    bool range_pending = false;                 // rvb_ir_time_range_begin has enqueued the HRTF time-range pass of the current configuration
    int hrtf_table_ears = 0;                    // ears of the HRTF table on the device: 2 after rvb_ir_configure_hrtf, 1 after the one-ear attenuate calls
    bool traced = false;
    uint64_t nreflections = 0;
    float mic[3] = {0, 0, 0};
    // last trace: npairs (source, microphone) pairs x nrays rays each; the IR stage works on one pair's slice at a time
    uint64_t npairs = 1, traced_rays = 0, ir_pair = 0;
    std::vector<float> pair_mics_host;          // [npairs][3]
    DevBuf pair_geom, pair_direct, pair_range;   // device: mics+sources [2*npairs] float4, direct [npairs] Impulse, ranges [npairs][2]
    void * pair_stage = nullptr;                 // pinned staging of the per-pair geometry of a launch
    size_t pair_stage_cap = 0;
    hipEvent_t pair_stage_free = nullptr;
    std::vector<rvb_impulse> pair_direct_host;
    std::vector<uint32_t> pair_range_host;
    DevBuf image_items;                          // work list of the image-source check kernel
    DevBuf impulses, early, candidates, small, stamps, sort_keys, sort_scratch, sort_order, group_temp;       // small: [0] candidate count, [2..3] executed, [16..] direct, range
    // host mirror of `small`, fetched once per trace together with the first few image-source candidates (usually all of them).
    // One PINNED block: a device-to-host copy into pageable memory is staged by the runtime and blocks the host per call (three
    // round trips of 30-160 us between the shadow kernel and the binning stage in a kernel trace); into pinned memory the copies
    // are asynchronous and the host waits once.
    unsigned char * small_host = nullptr;       // [kSmallBytes]
    rvb_image_candidate * first_candidates = nullptr;   // [kFirstCandidates], behind small_host in the same block
    uint32_t * range_host = nullptr;            // [2] behind them: where the HRTF time-range pass lands (rvb_ir_time_range_begin)
    bool small_valid = false;
    bool stamps_cleared = false;
    std::vector<Timing> timings;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    bool timing_failed = false;                 // an event could not be created: timings are dropped, the work itself is unaffected

    // impulse-response stage
    bool ir_configured = false;
    AttenuationModel model;
    int which = RVB_IR_ALL;
    DevBuf images, hrtf_table, acc, keys_a, keys_b, vals_a, vals_b, sort_temp, scratch_in, scratch_out, hist, bin_starts;
    DevBuf flat_in;                              // rvb_flatten's upload: a buffer of its own, so that no other entry point overwrites it between a size query and the fill
    DevBuf own_sort_temp, own_sort_keys, own_sort_values;      // csrc/radix_sort.hip: tile counters, the intermediate (key, value) pair
    uint64_t nimages = 0;
    std::vector<rvb_impulse> images_host;
    // exact mode in two steps (rvb_ir_exact_prepare / rvb_ir_exact_fold): what the sorted list in keys_b / vals_b / bin_starts was prepared for
    struct ExactState { bool valid = false; bool hrtf_combined = false; uint64_t nbins = 0, n = 0, ndiffuse = 0, nimages = 0; } exact;

    // staged host copies (rvb_copy_to_host / rvb_copy_to_device): per worker thread two pinned bounce buffers and a stream
    struct CopyLane { void * pinned[2] = {nullptr, nullptr}; hipStream_t stream = nullptr; hipEvent_t done[2] = {nullptr, nullptr}; };
    std::vector<CopyLane> copy_lanes;
    // rvb_flatten remembers what it uploaded for a size query, so that the fill that follows does not upload and key it again.
    // The array sits in flat_in (written by rvb_flatten only), its keys in keys_a / vals_a: every other writer of those two
    // (rvb_ir_accumulate in exact mode, rvb_flatten_device) clears flat_host, and so does a reallocation of the sort buffers.
    const void * flat_host = nullptr;
    uint64_t flat_n = 0, flat_bins = 0;
    float flat_rate = 0.0f;

    hipEvent_t next_event()
    {
        if (events_used == event_pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) { timing_failed = true; return nullptr; }
            event_pool.push_back(e);
        }
        return event_pool[events_used++];
    }
    void begin_timing(const char * name, hipStream_t on = nullptr)
    {
        Timing t;
        t.name = name;
        t.start = next_event();
        t.stop = next_event();
        timing_open = t.start && t.stop;
        if (!timing_open) return;
        (void) hipEventRecord(t.start, on ? on : stream);
        timings.push_back(t);
    }
    void end_timing(hipStream_t on = nullptr) { if (timing_open) (void) hipEventRecord(timings.back().stop, on ? on : stream); timing_open = false; }
    void reset_timings() { timings.clear(); events_used = 0; }
    bool timing_open = false;
};

static bool own_sort_enabled();
static int own_sort(rvb_ctx * ctx, const uint32_t * keys, uint32_t value_base, uint64_t n, int begin_bit, int end_bit,
                    uint32_t * keys_out, uint32_t * values_out, bool want_keys);

namespace {

int fail(rvb_ctx * ctx, int code, const std::string & what)
{
    if (ctx) ctx->error = what; else g_create_error = what;
    return code;
}

// the diffuse impulses the IR stage works on: the slice of the pair chosen with rvb_ir_select_pair (pair 0 of 1 otherwise)
const rvb_impulse * ir_diffuse(const rvb_ctx * ctx)
{
    return ctx->impulses.as<rvb_impulse>() + ctx->ir_pair * ctx->nrays * ctx->nreflections;
}

#define RVB_HIP(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(ctx, RVB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

#define RVB_BIND(ctx) RVB_HIP(ctx, hipSetDevice((ctx)->device))

// small-buffer layout (bytes)
const size_t kSmallCandidateCount = 0;
const size_t kSmallExecuted = 8;
const size_t kSmallRange = 16;      // two uint32
const size_t kSmallMaxTime = 24;    // uint32
const size_t kSmallTraceRange = 32; // two uint32: time range of the traced diffuse impulses (shadow_kernel)
const size_t kSmallImageItems = 40; // uint32: entries of the image-source work list (image_plan_kernel)
const size_t kSmallDirect = 64;     // rvb_impulse
const size_t kSmallBytes = 128;
const size_t kFirstCandidates = 32;

uint64_t bins_for(float max_time, float predelay, float sample_rate)
{
    const float t = max_time > predelay ? max_time - predelay : 0.0f;   // rayverb.h:89
    return (uint64_t) (roundf(t * sample_rate) + 1);                    // rayverb.cpp:57
}

int key_bits_for(uint64_t nbins)
{
    int bits = 1;
    while (bits < 32 && (1ull << bits) < nbins + 1)
        ++bits;
    return bits;
}

}  // namespace

extern "C" {

int rvb_create(rvb_ctx ** out, int device, unsigned flags)
{
    (void) flags;
    if (!out)
        return fail(nullptr, RVB_ERR_INVALID, "rvb_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, RVB_ERR_NO_DEVICE,
                    std::string("rvb_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                    "); this library has no CPU path");
    if (device < 0 || device >= count)
        return fail(nullptr, RVB_ERR_INVALID, "rvb_create: device index out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return fail(nullptr, RVB_ERR_NO_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RVB_ERR_NO_DEVICE, std::string("rvb_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    rvb_ctx * ctx = new rvb_ctx();
    ctx->device = device;
    ctx->arch = prop.gcnArchName;
    ctx->compute_units = prop.multiProcessorCount;
    ctx->hbm_bytes = prop.totalGlobalMem;
    // The side stream (image_kernel) runs at the lowest priority: the record grouping on the main stream is the critical
    // path between path_kernel and shadow_kernel and must not queue behind image_kernel's 14 k workgroups.
    int prio_least = 0, prio_greatest = 0;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, prio_least)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&ctx->export_stream, hipStreamNonBlocking, (prio_least + prio_greatest) / 2)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->export_ready, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->path_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->prep_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->group_done, hipEventDisableTiming)) != hipSuccess ||
        (e = ctx->small.ensure(kSmallBytes)) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void **>(&ctx->small_host), kSmallBytes + kFirstCandidates * sizeof(rvb_image_candidate) + 16, hipHostMallocDefault)) != hipSuccess) {
        std::string what = std::string("rvb_create: ") + hipGetErrorString(e);
        delete ctx;
        return fail(nullptr, RVB_ERR_HIP, what);
    }
    std::memset(ctx->small_host, 0, kSmallBytes + kFirstCandidates * sizeof(rvb_image_candidate) + 16);
    ctx->first_candidates = reinterpret_cast<rvb_image_candidate *>(ctx->small_host + kSmallBytes);
    ctx->range_host = reinterpret_cast<uint32_t *>(ctx->small_host + kSmallBytes + kFirstCandidates * sizeof(rvb_image_candidate));
    *out = ctx;
    return RVB_OK;
}

void rvb_destroy(rvb_ctx * ctx)
{
    if (!ctx)
        return;
    (void) hipSetDevice(ctx->device);
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    ctx->store.reset();
    for (DevBuf * b : {&ctx->sort_keys, &ctx->sort_scratch, &ctx->sort_order, &ctx->group_temp, &ctx->directions_own, &ctx->impulses, &ctx->image_items,
                       &ctx->early, &ctx->candidates, &ctx->small, &ctx->stamps, &ctx->images, &ctx->hrtf_table, &ctx->acc, &ctx->keys_a,
                       &ctx->keys_b, &ctx->vals_a, &ctx->vals_b, &ctx->sort_temp, &ctx->scratch_in, &ctx->scratch_out, &ctx->flat_in, &ctx->hist, &ctx->bin_starts, &ctx->own_sort_temp, &ctx->own_sort_keys, &ctx->own_sort_values,
                       &ctx->pair_geom, &ctx->pair_direct, &ctx->pair_range})
        b->release();
    for (rvb_ctx::CopyLane & l : ctx->copy_lanes) {
        for (int i = 0; i < 2; ++i) { if (l.pinned[i]) (void) hipHostFree(l.pinned[i]); if (l.done[i]) (void) hipEventDestroy(l.done[i]); }
        if (l.stream) (void) hipStreamDestroy(l.stream);
    }
    if (ctx->small_host) (void) hipHostFree(ctx->small_host);
    if (ctx->pair_stage) (void) hipHostFree(ctx->pair_stage);
    if (ctx->pair_stage_free) (void) hipEventDestroy(ctx->pair_stage_free);
    for (hipEvent_t e : ctx->event_pool) (void) hipEventDestroy(e);
    if (ctx->path_done) (void) hipEventDestroy(ctx->path_done);
    if (ctx->side_done) (void) hipEventDestroy(ctx->side_done);
    if (ctx->prep_done) (void) hipEventDestroy(ctx->prep_done);
    if (ctx->group_done) (void) hipEventDestroy(ctx->group_done);
    if (ctx->side_stream) { (void) hipStreamSynchronize(ctx->side_stream); (void) hipStreamDestroy(ctx->side_stream); }
    if (ctx->export_stream) { (void) hipStreamSynchronize(ctx->export_stream); (void) hipStreamDestroy(ctx->export_stream); }
    if (ctx->export_ready) (void) hipEventDestroy(ctx->export_ready);
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char * rvb_last_error(const rvb_ctx * ctx)
{
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

int rvb_wait_for_event(rvb_ctx * ctx, void * hip_event)
{
    if (!ctx || !hip_event) return RVB_ERR_INVALID;
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamWaitEvent(ctx->stream, reinterpret_cast<hipEvent_t>(hip_event), 0));
    return RVB_OK;
}

int rvb_record_event(rvb_ctx * ctx, void * hip_event)
{
    if (!ctx || !hip_event) return RVB_ERR_INVALID;
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipEventRecord(reinterpret_cast<hipEvent_t>(hip_event), ctx->stream));
    return RVB_OK;
}

int rvb_synchronize(rvb_ctx * ctx)
{
    if (!ctx) return RVB_ERR_INVALID;
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RVB_OK;
}

int rvb_device_info(rvb_ctx * ctx, char * arch, uint64_t arch_capacity, int * compute_units, uint64_t * hbm_bytes)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (arch && arch_capacity) {
        std::strncpy(arch, ctx->arch.c_str(), arch_capacity - 1);
        arch[arch_capacity - 1] = 0;
    }
    if (compute_units) *compute_units = ctx->compute_units;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return RVB_OK;
}

int rvb_device_index(rvb_ctx * ctx, int * device)
{
    if (!ctx || !device) return RVB_ERR_INVALID;
    *device = ctx->device;
    return RVB_OK;
}

int rvb_set_scene(rvb_ctx * ctx, const rvb_triangle * triangles, uint64_t ntriangles,
                  const rvb_float3 * vertices, uint64_t nvertices,
                  const rvb_surface * surfaces, uint64_t nsurfaces)
{
    if (!ctx) return RVB_ERR_INVALID;
    if ((ntriangles && !triangles) || (nvertices && !vertices) || !surfaces || nsurfaces == 0)
        return fail(ctx, RVB_ERR_INVALID, "rvb_set_scene: null input or no surfaces");
    RVB_BIND(ctx);
    BuiltScene built;
    std::string err = rvb_build_scene(triangles, ntriangles, vertices, nvertices, nsurfaces, built);
    if (!err.empty())
        return fail(ctx, err.find("stack") != std::string::npos ? RVB_ERR_CAPACITY : RVB_ERR_INVALID, "rvb_set_scene: " + err);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_scene = false;
    ctx->traced = false;
    // a store other contexts hold stays theirs: this context gets a new one (buffers it holds alone are reused)
    if (!ctx->store || ctx->store.use_count() > 1) {
        ctx->store = std::make_shared<SceneStore>();
        ctx->store->device = ctx->device;
    }
    SceneStore & st = *ctx->store;
    auto upload = [&](DevBuf & b, const void * src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(bytes);
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
    };
    RVB_HIP(ctx, upload(st.nodes, built.nodes.data(), built.nodes.size() * sizeof(BvhNode)));
    RVB_HIP(ctx, upload(st.tris, built.tris.data(), built.tris.size() * sizeof(BvhTri)));
    // the eighth word of a shading record (the builder's plane-group number, of no use on the device) carries the triangle's position
    // in leaf order: the path kernel's record-grouping key comes with the 32 bytes it reads anyway instead of from a gather of its own
    for (size_t i = 0; i < built.shade.size() && i < built.leafpos.size(); ++i) built.shade[i].group = built.leafpos[i];
    RVB_HIP(ctx, upload(st.shade, built.shade.data(), built.shade.size() * sizeof(TriShade)));
    RVB_HIP(ctx, upload(st.corners, built.corners.data(), built.corners.size() * sizeof(TriCorners)));
    RVB_HIP(ctx, upload(st.surfaces, surfaces, nsurfaces * sizeof(rvb_surface)));
    ctx->scene.nodes = st.nodes.as<const BvhNode>();
    ctx->scene.tris = st.tris.as<const BvhTri>();
    ctx->scene.shade = st.shade.as<const TriShade>();
    ctx->scene.corners = st.corners.as<const TriCorners>();
    ctx->scene.surfaces = st.surfaces.as<const rvb_surface>();
    ctx->scene.ntris = (uint32_t) built.tris.size();
    // cull slack along the ray: the float distance of a triangle may differ from the exact one
    ctx->scene.cull_abs = built.pad;
    ctx->scene.cull_rel = 1e-4f;
    ctx->nnodes = built.nodes.size();
    ctx->kept = built.tris.size();
    ctx->depth = built.depth;
    ctx->stack_need = built.stack_need;
    ctx->nsurfaces = nsurfaces;
    ctx->have_scene = true;
    return RVB_OK;
}

int rvb_share_scene(rvb_ctx * ctx, rvb_ctx * from)
{
    if (!ctx || !from) return RVB_ERR_INVALID;
    if (ctx == from) return RVB_OK;
    if (!from->have_scene || !from->store) return fail(ctx, RVB_ERR_STATE, "rvb_share_scene: the other context holds no scene");
    if (from->device != ctx->device) return fail(ctx, RVB_ERR_INVALID, "rvb_share_scene: the contexts are on different devices (a scene is shared within one GPU's memory)");
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));        // nothing of this context reads its old scene any more
    ctx->store = from->store;
    ctx->scene = from->scene;
    ctx->nnodes = from->nnodes;
    ctx->kept = from->kept;
    ctx->depth = from->depth;
    ctx->stack_need = from->stack_need;
    ctx->nsurfaces = from->nsurfaces;
    ctx->have_scene = true;
    ctx->traced = false;
    return RVB_OK;
}

int rvb_scene_info(rvb_ctx * ctx, uint64_t * nodes, uint64_t * kept_triangles, uint32_t * depth)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RVB_ERR_STATE, "rvb_scene_info: no scene");
    if (nodes) *nodes = ctx->nnodes;
    if (kept_triangles) *kept_triangles = ctx->kept;
    if (depth) *depth = ctx->depth;
    return RVB_OK;
}

int rvb_set_directions(rvb_ctx * ctx, const rvb_float3 * directions, uint64_t nrays)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (nrays && !directions) return fail(ctx, RVB_ERR_INVALID, "rvb_set_directions: null directions");
    // Unit vectors are the contract (reference getRandomDirections, helpers.cpp:63-81): distances, times and the diffuse cosine
    // are only meaningful for |d| = 1.  The pruning margins of the acceleration structure (box padding, cull slack, the
    // triangles dropped as unhittable at build time) hold for 0.5 <= |d| <= 2; anything outside, or not finite, is refused
    // rather than traced with weaker guarantees.
    for (uint64_t i = 0; i < nrays; ++i) {
        const float * d = directions[i].s;
        const float len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (!(len2 >= 0.25f && len2 <= 4.0f))
            return fail(ctx, RVB_ERR_INVALID, "rvb_set_directions: direction " + std::to_string(i) + " is not a unit vector (length^2 = " +
                                              std::to_string(len2) + "; 0.5 <= length <= 2 is accepted)");
    }
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RVB_HIP(ctx, ctx->directions_own.ensure(nrays * sizeof(rvb_float3)));
    if (nrays)
        RVB_HIP(ctx, hipMemcpy(ctx->directions_own.p, directions, nrays * sizeof(rvb_float3), hipMemcpyHostToDevice));
    ctx->directions = ctx->directions_own.as<const float4>();
    ctx->nrays = nrays;
    ctx->traced = false;
    return RVB_OK;
}

int rvb_set_directions_device(rvb_ctx * ctx, const void * d_directions, uint64_t nrays)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (nrays && !d_directions) return fail(ctx, RVB_ERR_INVALID, "rvb_set_directions_device: null directions");
    ctx->directions = reinterpret_cast<const float4 *>(d_directions);
    ctx->nrays = nrays;
    ctx->traced = false;
    return RVB_OK;
}

int rvb_set_concurrent_traces(rvb_ctx * ctx, uint32_t traces)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (traces == 0 || traces > (1u << 20)) return fail(ctx, RVB_ERR_INVALID, "rvb_set_concurrent_traces: 1 .. 2^20");
    ctx->concurrent_traces = traces;
    return RVB_OK;
}

int rvb_set_path_lanes(rvb_ctx * ctx, uint32_t lanes)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 4) return fail(ctx, RVB_ERR_INVALID, "rvb_set_path_lanes: 0 (automatic), 1, 2 or 4");
    ctx->path_lanes = lanes;
    return RVB_OK;
}

// A trace in three steps — buffers, fills and kernel arguments; the path kernel; everything after it — so that rvb_trace_group can put
// the path kernels of several contexts into ONE launch.
struct TracePlan {
    TraceArgs a;
    uint64_t npairs = 1, nrays = 0, nreflections = 0;
    int key_bits = 1;
};

static int trace_prepare(rvb_ctx * ctx, const float * mics, const float * sources, uint64_t npairs, uint64_t nreflections,
                         const float air_coefficient[8], uint64_t ray_offset, uint64_t rays_in_flight, TracePlan & plan)
{
    if (!ctx->have_scene) return fail(ctx, RVB_ERR_STATE, "rvb_trace: rvb_set_scene has not been called");
    if (!ctx->directions && ctx->nrays) return fail(ctx, RVB_ERR_STATE, "rvb_trace: no directions");
    if (nreflections >= (1ull << 31) || ctx->nrays * npairs * 9 >= (1ull << 32))
        return fail(ctx, RVB_ERR_CAPACITY, "rvb_trace: too many reflections or rays for one context");
    const float * mic = mics, * source = sources;
    RVB_BIND(ctx);
    ctx->traced = false;                              // (until trace_finish: a failure below must not leave the last trace's results half reset)
    const uint64_t nrays = ctx->nrays * npairs;       // rays of this launch
    const size_t imp_bytes = (size_t) nrays * nreflections * sizeof(rvb_impulse);
    const size_t early_bytes = (size_t) nrays * 9 * sizeof(uint32_t);
    RVB_HIP(ctx, ctx->impulses.ensure(imp_bytes));
    RVB_HIP(ctx, ctx->early.ensure(early_bytes));
    RVB_HIP(ctx, ctx->candidates.ensure((size_t) nrays * 9 * sizeof(rvb_image_candidate)));
    RVB_HIP(ctx, ctx->image_items.ensure(((size_t) nrays * 9 + npairs) * 3 * sizeof(uint32_t)));      // (ray, bounce) entries, then a state word each

    // reference rayverb.cpp:600-616: outputs start zero-filled — path_kernel writes every slot of the
    // impulse array itself (work record or zeros), so no 819 MB fill is needed here
    // (a probe that skips this 3.6 MB fill — the kernel trace of the pipeline shows it stretched to 0.7 ms beside a histogram's host copy, right in
    // front of the next path kernel — made the pipeline 2 % SLOWER, 4.52 -> 4.62 ms per IR, three alternating runs: the fills stay)
    if (early_bytes) RVB_HIP(ctx, hipMemsetAsync(ctx->early.p, 0xFF, early_bytes, ctx->stream));
    RVB_HIP(ctx, hipMemsetAsync(ctx->small.p, 0, kSmallBytes, ctx->stream));
    RVB_HIP(ctx, hipMemsetAsync(ctx->small.as<char>() + kSmallTraceRange, 0xFF, 4, ctx->stream));

    TraceArgs & a = plan.a;
    a.scene = ctx->scene;
    a.directions = ctx->directions;
    a.impulses = ctx->impulses.as<rvb_impulse>();
    a.early = ctx->early.as<uint32_t>();
    a.candidates = ctx->candidates.as<rvb_image_candidate>();
    a.candidate_count = reinterpret_cast<uint32_t *>(ctx->small.as<char>() + kSmallCandidateCount);
    a.image_items = reinterpret_cast<ImageItem *>(ctx->image_items.p);
    a.image_state = ctx->image_items.as<uint32_t>() + ((size_t) nrays * 9 + npairs) * 2;
    a.image_item_count = reinterpret_cast<uint32_t *>(ctx->small.as<char>() + kSmallImageItems);
    a.direct = reinterpret_cast<rvb_impulse *>(ctx->small.as<char>() + kSmallDirect);
    a.npairs = (uint32_t) npairs;
    a.rays_per_pair = (uint32_t) ctx->nrays;
    a.pair_mics = nullptr;
    a.pair_sources = nullptr;
    a.executed = reinterpret_cast<unsigned long long *>(ctx->small.as<char>() + kSmallExecuted);
    a.time_range = reinterpret_cast<uint32_t *>(ctx->small.as<char>() + kSmallTraceRange);
    if (npairs > 1) {
        // several pairs per launch: geometry, direct path and time range per pair live in arrays of their own
        // staged through pinned memory and copied in stream order: no host synchronisation in front of the launch (the staging
        // block is reused only after the copies of the previous launch have left it)
        const size_t geom_floats = 8 * npairs, init_words = 2 * npairs;
        const size_t stage_bytes = geom_floats * sizeof(float) + init_words * sizeof(uint32_t);
        if (stage_bytes > ctx->pair_stage_cap) {
            if (ctx->pair_stage) { RVB_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void) hipHostFree(ctx->pair_stage); ctx->pair_stage = nullptr; ctx->pair_stage_cap = 0; }
            RVB_HIP(ctx, hipHostMalloc(&ctx->pair_stage, stage_bytes, hipHostMallocDefault));
            ctx->pair_stage_cap = stage_bytes;
        }
        if (!ctx->pair_stage_free) RVB_HIP(ctx, hipEventCreateWithFlags(&ctx->pair_stage_free, hipEventDisableTiming));
        else RVB_HIP(ctx, hipEventSynchronize(ctx->pair_stage_free));
        float * geom = static_cast<float *>(ctx->pair_stage);
        uint32_t * init = reinterpret_cast<uint32_t *>(geom + geom_floats);
        std::memset(geom, 0, geom_floats * sizeof(float));
        for (uint64_t p = 0; p < npairs; ++p)
            for (int i = 0; i < 3; ++i) { geom[4 * p + i] = mics[3 * p + i]; geom[4 * (npairs + p) + i] = sources[3 * p + i]; }
        for (uint64_t p = 0; p < npairs; ++p) { init[2 * p] = 0xFFFFFFFFu; init[2 * p + 1] = 0u; }
        RVB_HIP(ctx, ctx->pair_geom.ensure(geom_floats * sizeof(float)));
        RVB_HIP(ctx, ctx->pair_direct.ensure(npairs * sizeof(rvb_impulse)));
        RVB_HIP(ctx, ctx->pair_range.ensure(npairs * 2 * sizeof(uint32_t)));
        RVB_HIP(ctx, hipMemcpyAsync(ctx->pair_geom.p, geom, geom_floats * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        RVB_HIP(ctx, hipMemcpyAsync(ctx->pair_range.p, init, init_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        RVB_HIP(ctx, hipEventRecord(ctx->pair_stage_free, ctx->stream));
        a.pair_mics = ctx->pair_geom.as<float4>();
        a.pair_sources = ctx->pair_geom.as<float4>() + npairs;
        a.direct = ctx->pair_direct.as<rvb_impulse>();
        a.time_range = ctx->pair_range.as<uint32_t>();
    }
    // record bucketing for coherent shadow rays (RVB_SHADOW_SORT=0 turns it off)
    static const bool sort_records = !(getenv("RVB_SHADOW_SORT") && getenv("RVB_SHADOW_SORT")[0] == '0');
    const uint64_t nrecords = nrays * nreflections;
    a.sort_keys = nullptr; a.sort_keys16 = nullptr; a.key_shift = 0; a.sort_order = nullptr;
    int key_bits = 1;
    while (key_bits < 32 && (1ull << key_bits) < ctx->scene.ntris) ++key_bits;
    // The grouping only has to bring neighbouring triangles together: the top 16 bits of the leaf position are two
    // onesweep passes instead of three (C2, 17 key bits: grouping 0.66 -> 0.51 ms beside image_kernel, shadow_kernel +0.02 ms).
    if (sort_records && nrecords && nrecords < (1ull << 32) && ctx->scene.ntris) {
        RVB_HIP(ctx, ctx->sort_keys.ensure(nrecords * 4));
        RVB_HIP(ctx, ctx->sort_scratch.ensure(nrecords * 4));
        RVB_HIP(ctx, ctx->sort_order.ensure(nrecords * 4));
        const size_t group_bytes = rvb_group_records_temp_bytes(ctx->nrays * nreflections);
        if (group_bytes == 0) return fail(ctx, RVB_ERR_HIP, "rvb_trace: radix sort size query failed");
        RVB_HIP(ctx, ctx->group_temp.ensure(group_bytes));
        // slots of escaped rays get key 0xFFFFFFFF from path_kernel: they land in the last bucket and the
        // shadow kernel skips them by their valid flag
        // 16-bit keys in 64-byte runs (trace_kernels.hip, flush_key_run) whenever a ray's row divides into whole runs and rocPRIM sorts;
        // 32-bit keys, one store per record, otherwise (and for RVB_SORT=own, RVB_KEY_RUNS=0)
        static const bool runs_off = getenv("RVB_KEY_RUNS") && getenv("RVB_KEY_RUNS")[0] == '0';
        if (nreflections % 32 == 0 && !own_sort_enabled() && !runs_off) {
            a.sort_keys16 = ctx->sort_keys.as<uint16_t>();
            a.key_shift = (uint32_t) std::max(0, key_bits - 16);
        } else {
            a.sort_keys = ctx->sort_keys.as<uint32_t>();
        }
    }
    a.nrays = nrays;
    a.nreflections = (uint32_t) nreflections;
    a.stack_entries = ctx->stack_need;
    a.lds_surfaces = rvb_lds_surfaces(ctx->stack_need, ctx->nsurfaces);
    a.scene_nodes = (uint32_t) ctx->nnodes;
    // (rays_in_flight: what a group launch carries in all; 0 = this trace alone, times the caller's hint)
    a.path_lanes = ctx->path_lanes ? ctx->path_lanes : (rays_in_flight ? rvb_path_lanes_for(rays_in_flight, 1) : rvb_path_lanes_for(nrays, ctx->concurrent_traces));
    a.ray_offset = ray_offset;
    for (int i = 0; i < 3; ++i) { a.mic[i] = mic[i]; a.source[i] = source[i]; ctx->mic[i] = mic[i]; }
    for (int i = 0; i < 8; ++i) a.air[i] = air_coefficient[i];

    // diagnostic builds (RVB_STAMPS): [0..15] path_kernel, [16..31] shadow_kernel
    RVB_HIP(ctx, ctx->stamps.ensure(32 * sizeof(unsigned long long)));
    // (zeroed per trace only where a diagnostic build may write them: one tiny fill kernel less on the stream of every shipped trace)
    static const bool stamps_on = getenv("RVB_STAMPS") != nullptr;
    if (stamps_on || !ctx->stamps_cleared) {
        RVB_HIP(ctx, hipMemsetAsync(ctx->stamps.p, 0, 32 * sizeof(unsigned long long), ctx->stream));
        ctx->stamps_cleared = true;
    }
    a.scene.stamps = ctx->stamps.as<unsigned long long>();

    // diagnostic only (timing probes whose path kernel leaves records unwritten, -DRVB_PROBE_NO_STORES): start from invalid records
    static const bool probe_zero = getenv("RVB_PROBE_ZERO_RECORDS") != nullptr;
    if (probe_zero) {
        RVB_HIP(ctx, hipMemsetAsync(ctx->impulses.p, 0, imp_bytes, ctx->stream));
        if (a.sort_keys || a.sort_keys16) RVB_HIP(ctx, hipMemsetAsync(ctx->sort_keys.p, 0xFF, nrecords * (a.sort_keys ? 4 : 2), ctx->stream));
    }

    ctx->reset_timings();
    plan.npairs = npairs;
    plan.nrays = nrays;
    plan.nreflections = nreflections;
    plan.key_bits = key_bits;
    return RVB_OK;
}

static int trace_finish(rvb_ctx * ctx, TracePlan & plan, const float * mics)
{
    TraceArgs & a = plan.a;
    const uint64_t npairs = plan.npairs, nrays = plan.nrays, nreflections = plan.nreflections;
    const int key_bits = plan.key_bits;
    static const int group_bits = getenv("RVB_SHADOW_SORT_BITS") ? atoi(getenv("RVB_SHADOW_SORT_BITS")) : 16;
    // image_kernel and the record grouping both depend on path_kernel only: the first (latency-bound) runs on the
    // side stream beside the second (bandwidth-bound); shadow_kernel, which rewrites the records image_kernel
    // reads, waits for both.
    RVB_HIP(ctx, hipEventRecord(ctx->path_done, ctx->stream));
    RVB_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->path_done, 0));
    ctx->begin_timing("image_kernel", ctx->side_stream);
    a.scene.stamps = nullptr;
    rvb_launch_images(a, ctx->side_stream);
    ctx->end_timing(ctx->side_stream);
    RVB_HIP(ctx, hipEventRecord(ctx->side_done, ctx->side_stream));
    a.scene.stamps = ctx->stamps.as<unsigned long long>() + 16;
    if (a.sort_keys || a.sort_keys16) {
        ctx->begin_timing("record_sort_kernels");
        a.sort_order = ctx->sort_order.as<uint32_t>();
        RVB_HIP(ctx, hipGetLastError());
        // one grouping per pair (a pair's shadow rays share a microphone; records are [pair][ray][bounce])
        const uint64_t per_pair = ctx->nrays * nreflections;
        for (uint64_t p = 0; p < npairs; ++p) {
            if (a.sort_keys16) {
                // key16 = leaf position >> key_shift: its top group_bits bits are bits [end - group_bits, end) with end = min(16, key_bits)
                const int end = std::min(16, key_bits);
                RVB_HIP(ctx, rvb_group_records16(ctx->group_temp.p, ctx->group_temp.cap, a.sort_keys16 + p * per_pair,
                                                 ctx->sort_scratch.as<uint16_t>() + p * per_pair, a.sort_order + p * per_pair, per_pair,
                                                 (uint32_t) (p * per_pair), std::max(0, end - group_bits), end, ctx->stream));
            } else if (own_sort_enabled()) {
                const int rc = own_sort(ctx, a.sort_keys + p * per_pair, (uint32_t) (p * per_pair), per_pair, std::max(0, key_bits - group_bits), key_bits,
                                        ctx->sort_scratch.as<uint32_t>() + p * per_pair, a.sort_order + p * per_pair, false);
                if (rc != RVB_OK) return rc;
            } else {
                RVB_HIP(ctx, rvb_group_records(ctx->group_temp.p, ctx->group_temp.cap, a.sort_keys + p * per_pair,
                                               ctx->sort_scratch.as<uint32_t>() + p * per_pair, a.sort_order + p * per_pair, per_pair,
                                               (uint32_t) (p * per_pair), std::max(0, key_bits - group_bits), key_bits, ctx->stream));
            }
        }
        ctx->end_timing();
    }
    RVB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
    ctx->begin_timing(rvb_shadow_lanes() == 2 ? "shadow_pair_kernel" : (rvb_shadow_lanes() == 1 ? "shadow_lane_kernel" : "shadow_kernel"));
    rvb_launch_shadow(a, ctx->stream);
    ctx->end_timing();
    RVB_HIP(ctx, hipGetLastError());
    ctx->nreflections = nreflections;
    ctx->traced = true;
    ctx->small_valid = false;
    ctx->ir_configured = false;
    ctx->exact.valid = false;
    ctx->npairs = npairs;
    ctx->traced_rays = nrays;
    ctx->ir_pair = 0;
    ctx->pair_mics_host.assign(mics, mics + 3 * npairs);
    return RVB_OK;
}

// the name a path launch is timed under: the kernel that ran (csrc/trace_kernels.hip, rvb_path_lanes_for)
static const char * path_kernel_name(uint32_t lanes) { return lanes == 1 ? "path_lane_kernel" : (lanes == 2 ? "path_pair_kernel" : "path_kernel"); }

static int trace_common(rvb_ctx * ctx, const float * mics, const float * sources, uint64_t npairs, uint64_t nreflections,
                        const float air_coefficient[8], uint64_t ray_offset)
{
    TracePlan plan;
    int rc = trace_prepare(ctx, mics, sources, npairs, nreflections, air_coefficient, ray_offset, 0, plan);
    if (rc != RVB_OK) return rc;
    ctx->begin_timing(path_kernel_name(plan.a.path_lanes));
    rvb_launch_path(plan.a, ctx->stream);
    ctx->end_timing();
    return trace_finish(ctx, plan, mics);
}

int rvb_trace_group(rvb_ctx ** ctxs, uint64_t count, const float * mics, const float * sources, uint64_t nreflections,
                    const float air_coefficient[8], const uint64_t * ray_offsets)
{
    if (!ctxs || count == 0 || count > RVB_MAX_GROUP) return RVB_ERR_INVALID;
    for (uint64_t i = 0; i < count; ++i)
        if (!ctxs[i]) return RVB_ERR_INVALID;
    if (!mics || !sources || !air_coefficient) return fail(ctxs[0], RVB_ERR_INVALID, "rvb_trace_group: null argument");
    for (uint64_t i = 0; i < count; ++i)
        for (uint64_t j = 0; j < i; ++j)
            if (ctxs[i] == ctxs[j]) return fail(ctxs[0], RVB_ERR_INVALID, "rvb_trace_group: a context is listed twice");
    uint64_t total_rays = 0;
    for (uint64_t i = 0; i < count; ++i) total_rays += ctxs[i]->nrays;
    TracePlan plans[RVB_MAX_GROUP];
    for (uint64_t i = 0; i < count; ++i) {
        const int rc = trace_prepare(ctxs[i], mics + 3 * i, sources + 3 * i, 1, nreflections, air_coefficient, ray_offsets ? ray_offsets[i] : 0,
                                     total_rays, plans[i]);
        if (rc != RVB_OK) return rc;
    }
    // one launch for all of them when they can share a kernel: one or two lanes per ray, one device, one LDS layout (stack depth, surfaces
    // staged, key runs or not: the launch's LDS is laid out once for all its workgroups)
    bool fused = count > 1 && plans[0].a.path_lanes <= 2;
    for (uint64_t i = 1; i < count && fused; ++i)
        fused = ctxs[i]->device == ctxs[0]->device && plans[i].a.path_lanes == plans[0].a.path_lanes && plans[i].a.stack_entries == plans[0].a.stack_entries
                && plans[i].a.lds_surfaces == plans[0].a.lds_surfaces && (plans[i].a.sort_keys16 != nullptr) == (plans[0].a.sort_keys16 != nullptr);
    for (uint64_t i = 0; i < count && fused; ++i) fused = plans[i].nrays > 0;
    if (fused) {
        rvb_ctx * lead = ctxs[0];
        RVB_BIND(lead);
        for (uint64_t i = 1; i < count; ++i) {                      // the others' fills come first
            RVB_HIP(ctxs[i], hipEventRecord(ctxs[i]->prep_done, ctxs[i]->stream));
            RVB_HIP(lead, hipStreamWaitEvent(lead->stream, ctxs[i]->prep_done, 0));
        }
        TraceArgs args[RVB_MAX_GROUP];
        for (uint64_t i = 0; i < count; ++i) args[i] = plans[i].a;
        lead->begin_timing(path_kernel_name(args[0].path_lanes));
        rvb_launch_path_group(args, (uint32_t) count, lead->stream);
        lead->end_timing();
        RVB_HIP(lead, hipEventRecord(lead->group_done, lead->stream));
        for (uint64_t i = 1; i < count; ++i) {
            ctxs[i]->begin_timing(path_kernel_name(args[0].path_lanes));   // (elapsed: from this stream's arrival to the end of the group's kernel)
            RVB_HIP(ctxs[i], hipStreamWaitEvent(ctxs[i]->stream, lead->group_done, 0));
            ctxs[i]->end_timing();
        }
    } else {
        for (uint64_t i = 0; i < count; ++i) {
            RVB_BIND(ctxs[i]);
            ctxs[i]->begin_timing(path_kernel_name(plans[i].a.path_lanes));
            rvb_launch_path(plans[i].a, ctxs[i]->stream);
            ctxs[i]->end_timing();
        }
    }
    for (uint64_t i = 0; i < count; ++i) {
        RVB_BIND(ctxs[i]);
        const int rc = trace_finish(ctxs[i], plans[i], mics + 3 * i);
        if (rc != RVB_OK) return rc;
    }
    return RVB_OK;
}

int rvb_trace(rvb_ctx * ctx, const float mic[3], const float source[3], uint64_t nreflections,
              const float air_coefficient[8], uint64_t ray_offset)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mic || !source || !air_coefficient) return fail(ctx, RVB_ERR_INVALID, "rvb_trace: null argument");
    return trace_common(ctx, mic, source, 1, nreflections, air_coefficient, ray_offset);
}

int rvb_trace_pairs(rvb_ctx * ctx, const float * mics, const float * sources, uint64_t npairs, uint64_t nreflections,
                    const float air_coefficient[8], uint64_t ray_offset)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mics || !sources || !air_coefficient || npairs == 0) return fail(ctx, RVB_ERR_INVALID, "rvb_trace_pairs: null argument or no pairs");
    return trace_common(ctx, mics, sources, npairs, nreflections, air_coefficient, ray_offset);
}

int rvb_ir_select_pair(rvb_ctx * ctx, uint64_t pair)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_ir_select_pair: nothing traced");
    if (pair >= ctx->npairs) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_select_pair: pair out of range");
    ctx->ir_pair = pair;
    ctx->ir_configured = false;
    for (int i = 0; i < 3; ++i) ctx->mic[i] = ctx->pair_mics_host[3 * pair + i];
    return RVB_OK;
}

// one synchronising 128-byte read per trace serves candidate count, direct path, time range, bounce count
static int fetch_small(rvb_ctx * ctx)
{
    if (ctx->small_valid)
        return RVB_OK;
    RVB_HIP(ctx, hipMemcpyAsync(ctx->small_host, ctx->small.p, kSmallBytes, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->traced_rays)      // capacity rays * 9 >= 32 entries unless there are fewer than 4 rays
        RVB_HIP(ctx, hipMemcpyAsync(ctx->first_candidates, ctx->candidates.p,
                                    std::min(kFirstCandidates * sizeof(rvb_image_candidate), (size_t) ctx->traced_rays * 9 * sizeof(rvb_image_candidate)),
                                    hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->npairs > 1) {     // per-pair direct paths and time ranges
        ctx->pair_direct_host.resize(ctx->npairs);
        ctx->pair_range_host.resize(2 * ctx->npairs);
        RVB_HIP(ctx, hipMemcpyAsync(ctx->pair_direct_host.data(), ctx->pair_direct.p, ctx->npairs * sizeof(rvb_impulse), hipMemcpyDeviceToHost, ctx->stream));
        RVB_HIP(ctx, hipMemcpyAsync(ctx->pair_range_host.data(), ctx->pair_range.p, ctx->npairs * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->small_valid = true;
    return RVB_OK;
}

int rvb_get_diffuse(rvb_ctx * ctx, rvb_impulse * out)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_get_diffuse: nothing traced");
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t bytes = (size_t) ctx->traced_rays * ctx->nreflections * sizeof(rvb_impulse);
    if (bytes) {
        if (!out) return fail(ctx, RVB_ERR_INVALID, "rvb_get_diffuse: null output");
        return rvb_copy_to_host(ctx, out, ctx->impulses.p, bytes);
    }
    return RVB_OK;
}

int rvb_diffuse_device(rvb_ctx * ctx, const void ** d_impulses, uint64_t * count)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_diffuse_device: nothing traced");
    if (d_impulses) *d_impulses = ctx->impulses.p;
    if (count) *count = ctx->traced_rays * ctx->nreflections;
    return RVB_OK;
}

int rvb_get_direct(rvb_ctx * ctx, rvb_impulse * out)
{
    if (!ctx || !out) return RVB_ERR_INVALID;
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_get_direct: nothing traced");
    RVB_BIND(ctx);
    int rc = fetch_small(ctx);
    if (rc != RVB_OK) return rc;
    if (ctx->npairs > 1) *out = ctx->pair_direct_host[ctx->ir_pair];       // of the pair chosen with rvb_ir_select_pair
    else std::memcpy(out, ctx->small_host + kSmallDirect, sizeof(rvb_impulse));
    return RVB_OK;
}

int rvb_get_image_candidates(rvb_ctx * ctx, rvb_image_candidate * out, uint64_t capacity, uint64_t * count)
{
    if (!ctx || !count) return RVB_ERR_INVALID;
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_get_image_candidates: nothing traced");
    RVB_BIND(ctx);
    int rc = fetch_small(ctx);
    if (rc != RVB_OK) return rc;
    uint32_t n = 0;
    std::memcpy(&n, ctx->small_host + kSmallCandidateCount, sizeof(n));
    *count = n;
    if (!out)
        return RVB_OK;                       // size query
    if (capacity < n)
        return fail(ctx, RVB_ERR_CAPACITY, "rvb_get_image_candidates: capacity too small");
    if (n) {
        if (n <= kFirstCandidates)
            std::memcpy(out, ctx->first_candidates, (size_t) n * sizeof(rvb_image_candidate));     // came with the small block
        else
            RVB_HIP(ctx, hipMemcpy(out, ctx->candidates.p, (size_t) n * sizeof(rvb_image_candidate), hipMemcpyDeviceToHost));
        std::sort(out, out + n, [](const rvb_image_candidate & x, const rvb_image_candidate & y) {
            return x.ray != y.ray ? x.ray < y.ray : x.slot < y.slot;
        });
    }
    return RVB_OK;
}

int rvb_merge_images(const rvb_image_candidate * candidates, uint64_t ncandidates,
                     const rvb_impulse * direct, int remove_direct,
                     rvb_impulse * out, uint64_t capacity, uint64_t * count)
{
    if (!count || (ncandidates && !candidates))
        return RVB_ERR_INVALID;
    // reference rayverb.cpp:654-676: for each ray j and k = 1..10 the key is the first k entries of
    // the ray's index row; inserted if absent when k == 1 or the last entry is non-zero.  Entries
    // are zero except where a candidate exists, so rows are rebuilt from the candidates alone.
    std::vector<rvb_image_candidate> sorted(candidates, candidates + ncandidates);
    std::sort(sorted.begin(), sorted.end(), [](const rvb_image_candidate & x, const rvb_image_candidate & y) {
        return x.ray != y.ray ? x.ray < y.ray : x.slot < y.slot;
    });
    std::map<std::vector<unsigned long>, rvb_impulse> tally;
    if (direct)
        tally[std::vector<unsigned long>(1, 0)] = *direct;      // k == 1: key {0} from ray 0
    size_t i = 0;
    while (i < sorted.size()) {
        size_t j = i;
        unsigned long row[RVB_NUM_IMAGE_SOURCE] = {0};
        while (j < sorted.size() && sorted[j].ray == sorted[i].ray) {
            if (sorted[j].slot == 0 || sorted[j].slot >= RVB_NUM_IMAGE_SOURCE)
                return RVB_ERR_INVALID;
            row[sorted[j].slot] = sorted[j].index;
            ++j;
        }
        for (size_t c = i; c < j; ++c) {
            std::vector<unsigned long> key(row, row + sorted[c].slot + 1);
            if (tally.find(key) == tally.end())
                tally[key] = sorted[c].impulse;
        }
        i = j;
    }
    if (remove_direct)
        tally.erase(std::vector<unsigned long>(1, 0));          // rayverb.cpp:695-696
    *count = tally.size();
    if (!out)
        return RVB_OK;
    if (capacity < tally.size())
        return RVB_ERR_CAPACITY;
    size_t w = 0;
    for (const auto & kv : tally)
        out[w++] = kv.second;
    return RVB_OK;
}

// ---- materialised attenuation / flatten ---------------------------------------------------------

static int upload_hrtf_table(rvb_ctx * ctx, const float * table, int ears)
{
    // device layout [ear][360*180 + 1][8]; the extra row is the zero padding behind quirk Q5
    const size_t row = 360 * 180;
    std::vector<float> padded((size_t) 2 * (row + 1) * 8, 0.0f);
    for (int e = 0; e < ears; ++e)
        std::memcpy(padded.data() + (size_t) e * (row + 1) * 8, table + (size_t) e * row * 8, row * 8 * sizeof(float));
    RVB_HIP(ctx, ctx->hrtf_table.ensure(padded.size() * sizeof(float)));
    RVB_HIP(ctx, hipMemcpy(ctx->hrtf_table.p, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
    return RVB_OK;
}

static int run_attenuate(rvb_ctx * ctx, const AttenuationModel & m, uint32_t channel, const rvb_impulse * in, uint64_t n,
                         rvb_attenuated_impulse * out)
{
    if (n == 0) return RVB_OK;
    if (!in || !out) return fail(ctx, RVB_ERR_INVALID, "attenuate: null buffer");
    RVB_HIP(ctx, ctx->scratch_in.ensure(n * sizeof(rvb_impulse)));
    RVB_HIP(ctx, ctx->scratch_out.ensure(n * sizeof(rvb_attenuated_impulse)));
    int rc = rvb_copy_to_device(ctx, ctx->scratch_in.p, in, n * sizeof(rvb_impulse));
    if (rc != RVB_OK) return rc;
    ctx->reset_timings();
    ctx->begin_timing("attenuate_kernel");
    rvb_launch_attenuate(m, channel, ctx->scratch_in.as<rvb_impulse>(), n, ctx->scratch_out.as<rvb_attenuated_impulse>(), ctx->stream);
    ctx->end_timing();
    RVB_HIP(ctx, hipGetLastError());
    return rvb_copy_to_host(ctx, out, ctx->scratch_out.p, n * sizeof(rvb_attenuated_impulse));
}

int rvb_attenuate_speaker(rvb_ctx * ctx, const float mic[3], const rvb_impulse * in, uint64_t n,
                          const rvb_speaker * speaker, rvb_attenuated_impulse * out)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mic || !speaker) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_speaker: null argument");
    RVB_BIND(ctx);
    AttenuationModel m;
    m.hrtf = 0;
    m.nchannels = 1;
    for (int i = 0; i < 3; ++i) { m.mic[i] = mic[i]; m.speaker_dir[0][i] = speaker->direction[i]; }
    m.speaker_coeff[0] = speaker->coefficient;
    return run_attenuate(ctx, m, 0, in, n, out);
}

int rvb_attenuate_speaker_device(rvb_ctx * ctx, const float mic[3], const void * d_in, uint64_t n,
                                 const rvb_speaker * speaker, void * d_out)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mic || !speaker) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_speaker_device: null argument");
    if (n && (!d_in || !d_out)) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_speaker_device: null buffer");
    if (n && d_in == d_out) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_speaker_device: in-place is not supported");
    RVB_BIND(ctx);
    AttenuationModel m;
    m.hrtf = 0;
    m.nchannels = 1;
    for (int i = 0; i < 3; ++i) { m.mic[i] = mic[i]; m.speaker_dir[0][i] = speaker->direction[i]; }
    m.speaker_coeff[0] = speaker->coefficient;
    ctx->reset_timings();
    ctx->begin_timing("attenuate_kernel");
    rvb_launch_attenuate(m, 0, reinterpret_cast<const rvb_impulse *>(d_in), n, reinterpret_cast<rvb_attenuated_impulse *>(d_out), ctx->stream);
    ctx->end_timing();
    RVB_HIP(ctx, hipGetLastError());
    return RVB_OK;
}

int rvb_attenuate_hrtf(rvb_ctx * ctx, const float mic[3], const rvb_impulse * in, uint64_t n,
                       const float * table, const float facing[3], const float up[3], uint64_t channel,
                       rvb_attenuated_impulse * out)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mic || !table || !facing || !up || channel > 1) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_hrtf: bad argument");
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // the table handed in is this ear's; park it in the ear's slot of the device image
    std::vector<float> both((size_t) 2 * 360 * 180 * 8, 0.0f);
    std::memcpy(both.data() + (size_t) channel * 360 * 180 * 8, table, (size_t) 360 * 180 * 8 * sizeof(float));
    int rc = upload_hrtf_table(ctx, both.data(), 2);
    if (rc != RVB_OK) return rc;
    ctx->hrtf_table_ears = 1;                 // (one ear's table in its slot: not what rvb_ir_configure_hrtf(table == NULL) may reuse)
    AttenuationModel m;
    m.hrtf = 1;
    m.nchannels = 2;
    m.hrtf_table = ctx->hrtf_table.as<const float>();
    for (int i = 0; i < 3; ++i) { m.mic[i] = mic[i]; m.facing[i] = facing[i]; m.up[i] = up[i]; }
    ctx->ir_configured = false;
    return run_attenuate(ctx, m, (uint32_t) channel, in, n, out);
}

int rvb_attenuate_hrtf_device(rvb_ctx * ctx, const float mic[3], const void * d_in, uint64_t n,
                              const float * table, const float facing[3], const float up[3], uint64_t channel, void * d_out)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mic || !table || !facing || !up || channel > 1) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_hrtf_device: bad argument");
    if (n && (!d_in || !d_out)) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_hrtf_device: null buffer");
    if (n && d_in == d_out) return fail(ctx, RVB_ERR_INVALID, "rvb_attenuate_hrtf_device: in-place is not supported");
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));           // the table image is about to be replaced
    std::vector<float> both((size_t) 2 * 360 * 180 * 8, 0.0f);
    std::memcpy(both.data() + (size_t) channel * 360 * 180 * 8, table, (size_t) 360 * 180 * 8 * sizeof(float));
    int rc = upload_hrtf_table(ctx, both.data(), 2);
    if (rc != RVB_OK) return rc;
    ctx->hrtf_table_ears = 1;                 // (one ear's table in its slot: not what rvb_ir_configure_hrtf(table == NULL) may reuse)
    AttenuationModel m;
    m.hrtf = 1;
    m.nchannels = 2;
    m.hrtf_table = ctx->hrtf_table.as<const float>();
    for (int i = 0; i < 3; ++i) { m.mic[i] = mic[i]; m.facing[i] = facing[i]; m.up[i] = up[i]; }
    ctx->ir_configured = false;
    ctx->reset_timings();
    ctx->begin_timing("attenuate_kernel");
    rvb_launch_attenuate(m, (uint32_t) channel, reinterpret_cast<const rvb_impulse *>(d_in), n, reinterpret_cast<rvb_attenuated_impulse *>(d_out), ctx->stream);
    ctx->end_timing();
    RVB_HIP(ctx, hipGetLastError());
    return RVB_OK;
}

// rocPRIM's radix sort unless RVB_SORT=own asks for the library's own (csrc/radix_sort.hip: same results, kernels that fit beside
// resident path waves; measured 3-4 % slower per IR in the bench pipeline, see the file's header).
static bool own_sort_enabled()
{
    static const bool own = getenv("RVB_SORT") && std::strcmp(getenv("RVB_SORT"), "own") == 0;
    return own;
}

// (keys[i], value_base + i) sorted on key bits [begin_bit, end_bit) into (keys_out, values_out); keys_out is always a buffer of n
// words (intermediate passes use it) but holds the sorted keys only if want_keys.
static int own_sort(rvb_ctx * ctx, const uint32_t * keys, uint32_t value_base, uint64_t n, int begin_bit, int end_bit,
                    uint32_t * keys_out, uint32_t * values_out, bool want_keys)
{
    if (n == 0 || end_bit <= begin_bit) return RVB_OK;
    const int passes = (end_bit - begin_bit + 7) / 8;
    RVB_HIP(ctx, ctx->own_sort_temp.ensure(rvb_radix_sort_temp_bytes(n)));
    uint32_t * tmp_k = nullptr, * tmp_v = nullptr;
    if (passes > 1) {
        RVB_HIP(ctx, ctx->own_sort_keys.ensure(n * 4));
        RVB_HIP(ctx, ctx->own_sort_values.ensure(n * 4));
        tmp_k = ctx->own_sort_keys.as<uint32_t>();
        tmp_v = ctx->own_sort_values.as<uint32_t>();
    }
    // passes alternate A, B, A, ...: the last one must land in the caller's buffers
    const bool last_in_b = ((passes - 1) & 1) != 0;
    uint32_t * ka = last_in_b ? tmp_k : keys_out, * va = last_in_b ? tmp_v : values_out;
    uint32_t * kb = last_in_b ? keys_out : tmp_k, * vb = last_in_b ? values_out : tmp_v;
    const uint32_t * ks = nullptr, * vs = nullptr;
    RVB_HIP(ctx, rvb_radix_sort_pairs(ctx->own_sort_temp.p, ctx->own_sort_temp.cap, keys, nullptr, value_base, ka, va, kb, vb, n,
                                      begin_bit, end_bit, want_keys, &ks, &vs, ctx->stream));
    if (vs != values_out || (want_keys && ks != keys_out)) return fail(ctx, RVB_ERR_HIP, "internal error: radix sort result in the wrong buffer");
    return RVB_OK;
}

static int ensure_sort_buffers(rvb_ctx * ctx, uint64_t n)
{
    if (n * 4 > ctx->keys_a.cap || n * 4 > ctx->vals_a.cap) ctx->flat_host = nullptr;      // the keys of a size query are about to be freed
    ctx->exact.valid = false;                 // (every caller rewrites the sort buffers)
    RVB_HIP(ctx, ctx->keys_a.ensure(n * 4));
    RVB_HIP(ctx, ctx->keys_b.ensure(n * 4));
    RVB_HIP(ctx, ctx->vals_a.ensure(n * 4));
    RVB_HIP(ctx, ctx->vals_b.ensure(n * 4));
    RVB_HIP(ctx, ctx->sort_temp.ensure(rvb_sort_temp_bytes(n)));
    return RVB_OK;
}

// keys + max time of a device-resident AttenuatedImpulse array -> *bins; then (out != NULL) sort, ordered sum, download
static int flatten_keys(rvb_ctx * ctx, const rvb_attenuated_impulse * d_in, uint64_t n, float sample_rate, uint64_t * bins)
{
    int rc = ensure_sort_buffers(ctx, n);
    if (rc != RVB_OK) return rc;
    uint32_t * max_bits = reinterpret_cast<uint32_t *>(ctx->small.as<char>() + kSmallMaxTime);
    RVB_HIP(ctx, hipMemsetAsync(max_bits, 0, 4, ctx->stream));
    rvb_launch_flat_keys(d_in, n, sample_rate, ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), max_bits, ctx->stream);
    uint32_t bits = 0;
    RVB_HIP(ctx, hipMemcpyAsync(&bits, max_bits, 4, hipMemcpyDeviceToHost, ctx->stream));
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float max_time;
    std::memcpy(&max_time, &bits, 4);
    *bins = bins_for(max_time, 0.0f, sample_rate);
    return RVB_OK;
}

static int flatten_sum(rvb_ctx * ctx, const rvb_attenuated_impulse * d_in, uint64_t n, uint64_t bins, float * out)
{
    RVB_HIP(ctx, ctx->hist.ensure(bins * 8 * sizeof(float)));
    if (own_sort_enabled()) {       // (the values the key pass wrote are the impulse indices 0 .. n-1: implicit)
        const int rc = own_sort(ctx, ctx->keys_a.as<uint32_t>(), 0u, n, 0, key_bits_for(bins), ctx->keys_b.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), true);
        if (rc != RVB_OK) return rc;
    } else {
        rvb_sort_pairs(ctx->sort_temp.p, ctx->sort_temp.cap, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(),
                       ctx->vals_a.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), n, key_bits_for(bins), ctx->stream);
    }
    RVB_HIP(ctx, ctx->bin_starts.ensure(bins * 8));
    RVB_HIP(ctx, hipMemsetAsync(ctx->bin_starts.p, 0xFF, bins * 4, ctx->stream));
    rvb_launch_bin_starts(ctx->keys_b.as<uint32_t>(), n, bins, ctx->bin_starts.as<uint32_t>(), ctx->bin_starts.as<uint32_t>() + bins, ctx->stream);
    rvb_launch_flat_ordered_sum(d_in, ctx->keys_b.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), ctx->bin_starts.as<uint32_t>(), n, bins,
                                ctx->hist.as<float>(), ctx->stream);
    RVB_HIP(ctx, hipGetLastError());
    RVB_HIP(ctx, hipMemcpyAsync(out, ctx->hist.p, bins * 8 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RVB_OK;
}

int rvb_flatten(rvb_ctx * ctx, const rvb_attenuated_impulse * in, uint64_t n, float sample_rate,
                float * out, uint64_t capacity_bins, uint64_t * nbins)
{
    if (!ctx || !nbins) return RVB_ERR_INVALID;
    if (n && !in) return fail(ctx, RVB_ERR_INVALID, "rvb_flatten: null input");
    if (n >= (1ull << 32)) return fail(ctx, RVB_ERR_CAPACITY, "rvb_flatten: too many impulses");
    RVB_BIND(ctx);
    // the fill that follows a size query of the same array finds it (and its keys) on the device
    const bool resident = out && ctx->flat_host == in && ctx->flat_n == n && ctx->flat_rate == sample_rate && in != nullptr;
    uint64_t bins = ctx->flat_bins;
    if (!resident) {
        ctx->flat_host = nullptr;
        RVB_HIP(ctx, ctx->flat_in.ensure(n * sizeof(rvb_attenuated_impulse)));
        if (n) {
            int rc = rvb_copy_to_device(ctx, ctx->flat_in.p, in, n * sizeof(rvb_attenuated_impulse));
            if (rc != RVB_OK) return rc;
        }
        int rc = flatten_keys(ctx, ctx->flat_in.as<rvb_attenuated_impulse>(), n, sample_rate, &bins);
        if (rc != RVB_OK) return rc;
    }
    *nbins = bins;
    if (!out) {
        ctx->flat_host = in; ctx->flat_n = n; ctx->flat_rate = sample_rate; ctx->flat_bins = bins;
        return RVB_OK;
    }
    ctx->flat_host = nullptr;                 // (the sort consumes the keys)
    if (capacity_bins < bins)
        return fail(ctx, RVB_ERR_CAPACITY, "rvb_flatten: capacity_bins too small");
    return flatten_sum(ctx, ctx->flat_in.as<rvb_attenuated_impulse>(), n, bins, out);
}

int rvb_flatten_device(rvb_ctx * ctx, const void * d_attenuated, uint64_t n, float sample_rate,
                       float * out, uint64_t capacity_bins, uint64_t * nbins)
{
    if (!ctx || !nbins) return RVB_ERR_INVALID;
    if (n && !d_attenuated) return fail(ctx, RVB_ERR_INVALID, "rvb_flatten_device: null input");
    if (n >= (1ull << 32)) return fail(ctx, RVB_ERR_CAPACITY, "rvb_flatten_device: too many impulses");
    RVB_BIND(ctx);
    ctx->flat_host = nullptr;
    uint64_t bins = 0;
    int rc = flatten_keys(ctx, reinterpret_cast<const rvb_attenuated_impulse *>(d_attenuated), n, sample_rate, &bins);
    if (rc != RVB_OK) return rc;
    *nbins = bins;
    if (!out)
        return RVB_OK;
    if (capacity_bins < bins)
        return fail(ctx, RVB_ERR_CAPACITY, "rvb_flatten_device: capacity_bins too small");
    return flatten_sum(ctx, reinterpret_cast<const rvb_attenuated_impulse *>(d_attenuated), n, bins, out);
}

int rvb_fix_predelay_device(rvb_ctx * ctx, void * d_attenuated, uint64_t n, float seconds)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (n && !d_attenuated) return fail(ctx, RVB_ERR_INVALID, "rvb_fix_predelay_device: null array");
    RVB_BIND(ctx);
    rvb_launch_fix_predelay(reinterpret_cast<rvb_attenuated_impulse *>(d_attenuated), n, seconds, ctx->stream);
    RVB_HIP(ctx, hipGetLastError());
    return RVB_OK;
}

// ---- device buffers and staged host copies ----------------------------------------------------------

int rvb_device_alloc(rvb_ctx * ctx, uint64_t bytes, void ** d_ptr)
{
    if (!ctx || !d_ptr) return RVB_ERR_INVALID;
    RVB_BIND(ctx);
    *d_ptr = nullptr;
    RVB_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 16));
    return RVB_OK;
}

int rvb_device_free(rvb_ctx * ctx, void * d_ptr)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!d_ptr) return RVB_OK;
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RVB_HIP(ctx, hipStreamSynchronize(ctx->export_stream));       // ... or a copy out of it
    RVB_HIP(ctx, hipFree(d_ptr));
    return RVB_OK;
}

namespace {

const size_t kCopyChunk = 8u << 20;          // bytes per pinned bounce buffer

int copy_lane_count()
{
    static const int lanes = [] {
        if (const char * e = getenv("RVB_COPY_THREADS")) return std::max(1, std::min(32, atoi(e)));
        const unsigned hw = std::thread::hardware_concurrency();
        return (int) std::max(1u, std::min(8u, hw ? hw / 2 : 4u));
    }();
    return lanes;
}

hipError_t ensure_copy_lanes(rvb_ctx * ctx)
{
    if (!ctx->copy_lanes.empty()) return hipSuccess;
    std::vector<rvb_ctx::CopyLane> lanes((size_t) copy_lane_count());
    for (rvb_ctx::CopyLane & l : lanes) {
        hipError_t e;
        for (int i = 0; i < 2; ++i) {
            if ((e = hipHostMalloc(&l.pinned[i], kCopyChunk, hipHostMallocDefault)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&l.done[i], hipEventDisableTiming)) != hipSuccess) return e;
        }
        if ((e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking)) != hipSuccess) return e;
    }
    ctx->copy_lanes.swap(lanes);
    return hipSuccess;
}

// One lane's slice: chunk k+1 is on the link while chunk k is copied between the bounce buffer and pageable memory.
hipError_t lane_copy(int device, rvb_ctx::CopyLane & l, char * host, char * dev, size_t bytes, bool to_host)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    const size_t nchunks = (bytes + kCopyChunk - 1) / kCopyChunk;
    auto span = [&](size_t k) { return std::min(kCopyChunk, bytes - k * kCopyChunk); };
    if (to_host) {
        if (nchunks && (e = hipMemcpyAsync(l.pinned[0], dev, span(0), hipMemcpyDeviceToHost, l.stream)) != hipSuccess) return e;
        if (nchunks && (e = hipEventRecord(l.done[0], l.stream)) != hipSuccess) return e;
        for (size_t k = 0; k < nchunks; ++k) {
            if (k + 1 < nchunks) {
                if ((e = hipMemcpyAsync(l.pinned[(k + 1) & 1], dev + (k + 1) * kCopyChunk, span(k + 1), hipMemcpyDeviceToHost, l.stream)) != hipSuccess) return e;
                if ((e = hipEventRecord(l.done[(k + 1) & 1], l.stream)) != hipSuccess) return e;
            }
            if ((e = hipEventSynchronize(l.done[k & 1])) != hipSuccess) return e;
            std::memcpy(host + k * kCopyChunk, l.pinned[k & 1], span(k));
        }
    } else {
        for (size_t k = 0; k < nchunks; ++k) {
            if (k >= 2 && (e = hipEventSynchronize(l.done[k & 1])) != hipSuccess) return e;      // the buffer's previous chunk has left
            std::memcpy(l.pinned[k & 1], host + k * kCopyChunk, span(k));
            if ((e = hipMemcpyAsync(dev + k * kCopyChunk, l.pinned[k & 1], span(k), hipMemcpyHostToDevice, l.stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(l.done[k & 1], l.stream)) != hipSuccess) return e;
        }
        if ((e = hipStreamSynchronize(l.stream)) != hipSuccess) return e;
    }
    return hipSuccess;
}

int staged_copy(rvb_ctx * ctx, void * host, void * dev, uint64_t bytes, bool to_host)
{
    if (bytes == 0) return RVB_OK;
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));          // ordered after the context's work
    if (bytes < (4u << 20)) {                                 // small: one plain copy
        RVB_HIP(ctx, to_host ? hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) : hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        return RVB_OK;
    }
    RVB_HIP(ctx, ensure_copy_lanes(ctx));
    const size_t lanes = ctx->copy_lanes.size();
    // slices are multiples of the chunk so that every lane moves whole chunks (2 MiB-aligned destinations keep the page
    // faults of fresh memory apart)
    const size_t chunks = (bytes + kCopyChunk - 1) / kCopyChunk, per = (chunks + lanes - 1) / lanes;
    std::vector<hipError_t> status(lanes, hipSuccess);
    std::vector<std::thread> workers;
    for (size_t i = 0; i < lanes; ++i) {
        const size_t first = i * per * kCopyChunk;
        if (first >= bytes) break;
        const size_t len = std::min((size_t) bytes - first, per * kCopyChunk);
        workers.emplace_back([=, &status] {
            status[i] = lane_copy(ctx->device, ctx->copy_lanes[i], static_cast<char *>(host) + first, static_cast<char *>(dev) + first, len, to_host);
        });
    }
    for (std::thread & t : workers) t.join();
    for (hipError_t e : status)
        if (e != hipSuccess) return fail(ctx, RVB_ERR_HIP, std::string("staged copy: ") + hipGetErrorString(e));
    return RVB_OK;
}

}  // namespace

int rvb_copy_to_host(rvb_ctx * ctx, void * dst, const void * d_src, uint64_t bytes)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (bytes && (!dst || !d_src)) return fail(ctx, RVB_ERR_INVALID, "rvb_copy_to_host: null pointer");
    RVB_BIND(ctx);
    return staged_copy(ctx, dst, const_cast<void *>(d_src), bytes, true);
}

int rvb_copy_to_device(rvb_ctx * ctx, void * d_dst, const void * src, uint64_t bytes)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (bytes && (!d_dst || !src)) return fail(ctx, RVB_ERR_INVALID, "rvb_copy_to_device: null pointer");
    RVB_BIND(ctx);
    return staged_copy(ctx, const_cast<void *>(src), d_dst, bytes, false);
}

int rvb_host_alloc(rvb_ctx * ctx, uint64_t bytes, void ** host_ptr)
{
    if (!ctx || !host_ptr) return RVB_ERR_INVALID;
    RVB_BIND(ctx);
    *host_ptr = nullptr;
    RVB_HIP(ctx, hipHostMalloc(host_ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return RVB_OK;
}

int rvb_host_free(rvb_ctx * ctx, void * host_ptr)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!host_ptr) return RVB_OK;
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RVB_HIP(ctx, hipStreamSynchronize(ctx->export_stream));       // a copy into this block may still be on its way
    RVB_HIP(ctx, hipHostFree(host_ptr));
    return RVB_OK;
}

int rvb_copy_to_pinned_host_async(rvb_ctx * ctx, void * pinned_dst, const void * d_src, uint64_t bytes)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (bytes && (!pinned_dst || !d_src)) return fail(ctx, RVB_ERR_INVALID, "rvb_copy_to_pinned_host_async: null pointer");
    if (bytes == 0) return RVB_OK;
    RVB_BIND(ctx);
    // Ordered behind what the context's stream holds now, but on a stream of its own: neither the host nor the context's next trace
    // waits for the link.  The copy itself is the runtime's (a blit kernel on this box: no DMA engine takes it).  Measured at
    // workload C2, ms per IR in the bench pipeline (profiles/r03_export_variants_n1.txt): no copy 4.51-4.79, this 4.57-4.62, the same
    // copy issued from a torch side stream when the IR is handed over 4.96, in stream order on the context's own stream 5.90, and
    // copy kernels of this library's own with 2-512 waves and plain / nt / sc1 / sc0 sc1 stores 4.83-5.97 — stores to host memory
    // from a few long-lived waves slow every other kernel's memory traffic down for as long as they last.
    RVB_HIP(ctx, hipEventRecord(ctx->export_ready, ctx->stream));
    RVB_HIP(ctx, hipStreamWaitEvent(ctx->export_stream, ctx->export_ready, 0));
    RVB_HIP(ctx, hipMemcpyAsync(pinned_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->export_stream));
    return RVB_OK;
}

int rvb_synchronize_exports(rvb_ctx * ctx)
{
    if (!ctx) return RVB_ERR_INVALID;
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->export_stream));
    return RVB_OK;
}

// ---- fused impulse-response stage ----------------------------------------------------------------

static int configure_common(rvb_ctx * ctx, int which, const rvb_impulse * images, uint64_t nimages)
{
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_ir_configure: nothing traced");
    if (which < 1 || which > 3) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_configure: which must be 1..3");
    if (nimages && !images) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_configure: null images");
    if (nimages * sizeof(rvb_impulse) > ctx->images.cap)
        RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the buffer is about to be replaced
    RVB_HIP(ctx, ctx->images.ensure(nimages * sizeof(rvb_impulse)));
    ctx->nimages = nimages;
    ctx->images_host.assign(images, images + nimages);
    // in stream order (kernels of an earlier configuration that read the old images run before it); the source is the
    // context's own copy, which lives until the next configure
    if (nimages) RVB_HIP(ctx, hipMemcpyAsync(ctx->images.p, ctx->images_host.data(), nimages * sizeof(rvb_impulse), hipMemcpyHostToDevice, ctx->stream));
    ctx->which = which;
    ctx->ir_configured = true;
    ctx->exact.valid = false;
    ctx->range_pending = false;
    return RVB_OK;
}

int rvb_ir_configure_speakers(rvb_ctx * ctx, const float mic[3], const rvb_speaker * speakers, uint64_t nspeakers,
                              int which, const rvb_impulse * images, uint64_t nimages)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mic || !speakers || nspeakers == 0 || nspeakers > 8)
        return fail(ctx, RVB_ERR_INVALID, "rvb_ir_configure_speakers: 1..8 speakers required");
    RVB_BIND(ctx);
    AttenuationModel m;
    m.hrtf = 0;
    m.nchannels = (uint32_t) nspeakers;
    for (int i = 0; i < 3; ++i) m.mic[i] = mic[i];
    for (uint64_t s = 0; s < nspeakers; ++s) {
        for (int i = 0; i < 3; ++i) m.speaker_dir[s][i] = speakers[s].direction[i];
        m.speaker_coeff[s] = speakers[s].coefficient;
    }
    ctx->model = m;
    return configure_common(ctx, which, images, nimages);
}

int rvb_ir_configure_hrtf(rvb_ctx * ctx, const float mic[3], const float * table, const float facing[3], const float up[3],
                          int which, const rvb_impulse * images, uint64_t nimages)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!mic || !facing || !up) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_configure_hrtf: null argument");
    RVB_BIND(ctx);
    if (table) {
        // (the table on the device may still be read by kernels enqueued under the previous configuration)
        RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int rc = upload_hrtf_table(ctx, table, 2);
        if (rc != RVB_OK) return rc;
        ctx->hrtf_table_ears = 2;
    } else if (ctx->hrtf_table_ears != 2) {
        // table == NULL: the two-ear table of the previous rvb_ir_configure_hrtf on this context stays (a caller that configures many
        // listeners with one table — the pipeline — uploads its 4 MB once, not per impulse response)
        return fail(ctx, RVB_ERR_STATE, "rvb_ir_configure_hrtf: table == NULL needs an earlier call with a table on this context");
    }
    AttenuationModel m;
    m.hrtf = 1;
    m.nchannels = 2;
    m.hrtf_table = ctx->hrtf_table.as<const float>();
    for (int i = 0; i < 3; ++i) { m.mic[i] = mic[i]; m.facing[i] = facing[i]; m.up[i] = up[i]; }
    ctx->model = m;
    return configure_common(ctx, which, images, nimages);
}

// HRTF model: the attenuated time of an impulse differs per ear (kernel.cpp:616-622), so the range needs a pass over the impulses; the
// pass and the copy of its two words to pinned host memory are ENQUEUED here and waited for in rvb_ir_time_range — a caller with several
// contexts enqueues all of them (rvb_ir_time_range_begin) before it waits for the first.
static int time_range_enqueue(rvb_ctx * ctx)
{
    uint32_t * range = reinterpret_cast<uint32_t *>(ctx->small.as<char>() + kSmallRange);
    RVB_HIP(ctx, hipMemsetAsync(range, 0xFF, 4, ctx->stream));
    RVB_HIP(ctx, hipMemsetAsync(range + 1, 0, 4, ctx->stream));
    ctx->reset_timings();
    ctx->begin_timing("time_range_kernel");
    if (ctx->which & RVB_IR_DIFFUSE)
        rvb_launch_time_range(ctx->model, ir_diffuse(ctx), ctx->nrays * ctx->nreflections, range, ctx->stream);
    if (ctx->which & RVB_IR_IMAGES)
        rvb_launch_time_range(ctx->model, ctx->images.as<rvb_impulse>(), ctx->nimages, range, ctx->stream);
    ctx->end_timing();
    RVB_HIP(ctx, hipGetLastError());
    RVB_HIP(ctx, hipMemcpyAsync(ctx->range_host, range, 8, hipMemcpyDeviceToHost, ctx->stream));
    ctx->range_pending = true;
    return RVB_OK;
}

int rvb_ir_time_range_begin(rvb_ctx * ctx)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->ir_configured) return fail(ctx, RVB_ERR_STATE, "rvb_ir_time_range_begin: rvb_ir_configure_* first");
    RVB_BIND(ctx);
    if (!ctx->model.hrtf) return RVB_OK;          // speakers: the range came with the trace (reduced inside the shadow kernel)
    return time_range_enqueue(ctx);
}

int rvb_ir_time_range(rvb_ctx * ctx, float * min_nonzero_time, float * max_time)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->ir_configured) return fail(ctx, RVB_ERR_STATE, "rvb_ir_time_range: rvb_ir_configure_* first");
    RVB_BIND(ctx);
    if (!ctx->model.hrtf) {
        // Speaker channels keep the input time (kernel.cpp:530-533), so the range is that of the raw
        // impulses: the diffuse part was reduced inside shadow_kernel, the few images are scanned here.
        ctx->reset_timings();
        uint32_t got[2] = {0xFFFFFFFFu, 0u};
        if (ctx->which & RVB_IR_DIFFUSE) {
            int rc = fetch_small(ctx);
            if (rc != RVB_OK) return rc;
            if (ctx->npairs > 1) { got[0] = ctx->pair_range_host[2 * ctx->ir_pair]; got[1] = ctx->pair_range_host[2 * ctx->ir_pair + 1]; }
            else std::memcpy(got, ctx->small_host + kSmallTraceRange, sizeof(got));
        }
        float lo = 0.0f, hi = 0.0f;
        bool have_lo = got[0] != 0xFFFFFFFFu;
        if (have_lo) std::memcpy(&lo, &got[0], 4);
        std::memcpy(&hi, &got[1], 4);
        if (ctx->which & RVB_IR_IMAGES)
            for (const rvb_impulse & im : ctx->images_host) {
                bool nonzero = false;
                for (int b = 0; b < 8; ++b) nonzero = nonzero || im.volume[b] != 0.0f;
                if (!nonzero) continue;
                if (im.time != 0.0f && (!have_lo || im.time < lo)) { lo = im.time; have_lo = true; }
                if (im.time > hi) hi = im.time;
            }
        if (min_nonzero_time) *min_nonzero_time = have_lo ? lo : 0.0f;
        if (max_time) *max_time = hi;
        return RVB_OK;
    }
    if (!ctx->range_pending) {
        const int rc = time_range_enqueue(ctx);
        if (rc != RVB_OK) return rc;
    }
    ctx->range_pending = false;
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t got[2];
    std::memcpy(got, ctx->range_host, sizeof(got));
    float lo = 0.0f, hi;
    if (got[0] != 0xFFFFFFFFu) std::memcpy(&lo, &got[0], 4);
    std::memcpy(&hi, &got[1], 4);
    if (min_nonzero_time) *min_nonzero_time = lo;
    if (max_time) *max_time = hi;
    return RVB_OK;
}

uint64_t rvb_ir_bins(float max_time, float predelay, float sample_rate)
{
    return bins_for(max_time, predelay, sample_rate);
}

// Exact mode, step 1 (everything that does not depend on what the histogram holds): per-impulse bin keys, the radix sort, where each
// bin's run starts and ends.  Speaker channels keep the input time (kernel.cpp:530-533) and share ONE sorted list; the two HRTF ears
// shift the arrival time differently (kernel.cpp:616-622) and are keyed in one pass into ONE list of 2 n entries (bin_keys_hrtf_kernel).
static int exact_prepare(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins)
{
    const AttenuationModel & m = ctx->model;
    const uint64_t ndiffuse = (ctx->which & RVB_IR_DIFFUSE) ? ctx->nrays * ctx->nreflections : 0;
    const uint64_t nimages = (ctx->which & RVB_IR_IMAGES) ? ctx->nimages : 0;
    const uint64_t n = ndiffuse + nimages;
    if (n >= (1ull << 31)) return fail(ctx, RVB_ERR_CAPACITY, "rvb_ir_accumulate: too many impulses for exact mode");
    if (nbins >= 0x7FFFFFF0ull) return fail(ctx, RVB_ERR_CAPACITY, "rvb_ir_accumulate: too many bins for exact mode");
    ctx->flat_host = nullptr;                 // keys_a / vals_a are rewritten below: a pending rvb_flatten size query is void
    if (m.hrtf) {
        int rc = ensure_sort_buffers(ctx, 2 * n);
        if (rc != RVB_OK) return rc;
        const uint64_t nkeys = 2 * (nbins + 1);
        const int bits = key_bits_for(nkeys - 1);
        RVB_HIP(ctx, ctx->bin_starts.ensure(nkeys * 8));       // starts, then ends
        rvb_launch_bin_keys_hrtf(m, ir_diffuse(ctx), ndiffuse, 0, n, predelay, sample_rate, (uint32_t) nbins,
                                 ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), ctx->stream);
        rvb_launch_bin_keys_hrtf(m, ctx->images.as<rvb_impulse>(), nimages, ndiffuse, n, predelay, sample_rate, (uint32_t) nbins,
                                 ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), ctx->stream);
        // (explicit values — the two halves carry the same impulse numbers — and always rocPRIM's sort: RVB_SORT=own takes identity values only.
        // Values derived from the entry's position by a transform iterator instead of an array: 1.52 -> 1.58 ms, and 0.85 -> 0.88 ms for the
        // one-list speaker form; rocPRIM's first pass reads an array faster than it evaluates an iterator.)
        rvb_sort_pairs(ctx->sort_temp.p, ctx->sort_temp.cap, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(),
                       ctx->vals_a.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), 2 * n, bits, ctx->stream);
        RVB_HIP(ctx, hipMemsetAsync(ctx->bin_starts.p, 0xFF, nkeys * 4, ctx->stream));
        rvb_launch_bin_starts(ctx->keys_b.as<uint32_t>(), 2 * n, nkeys, ctx->bin_starts.as<uint32_t>(), ctx->bin_starts.as<uint32_t>() + nkeys, ctx->stream);
    } else {
        int rc = ensure_sort_buffers(ctx, n);
        if (rc != RVB_OK) return rc;
        // keys are bins, nbins itself marks "adds nothing": key_bits_for(nbins) bits cover 0 .. nbins
        const int bits = key_bits_for(nbins);
        const uint32_t sentinel = (uint32_t) nbins;
        RVB_HIP(ctx, ctx->bin_starts.ensure(nbins * 8));          // starts, then ends
        rvb_launch_bin_keys(m, 0, ir_diffuse(ctx), ndiffuse, 0, predelay, sample_rate, sentinel,
                            ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), ctx->stream);
        rvb_launch_bin_keys(m, 0, ctx->images.as<rvb_impulse>(), nimages, ndiffuse, predelay, sample_rate, sentinel,
                            ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), ctx->stream);
        if (own_sort_enabled()) {
            const int rc2 = own_sort(ctx, ctx->keys_a.as<uint32_t>(), 0u, n, 0, bits, ctx->keys_b.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), true);
            if (rc2 != RVB_OK) return rc2;
        } else {
            rvb_sort_pairs(ctx->sort_temp.p, ctx->sort_temp.cap, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(),
                           ctx->vals_a.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), n, bits, ctx->stream);
        }
        RVB_HIP(ctx, hipMemsetAsync(ctx->bin_starts.p, 0xFF, nbins * 4, ctx->stream));
        rvb_launch_bin_starts(ctx->keys_b.as<uint32_t>(), n, nbins, ctx->bin_starts.as<uint32_t>(), ctx->bin_starts.as<uint32_t>() + nbins, ctx->stream);
    }
    RVB_HIP(ctx, hipGetLastError());
    ctx->exact.valid = true;                  // (ensure_sort_buffers above cleared it)
    ctx->exact.hrtf_combined = m.hrtf != 0;
    ctx->exact.nbins = nbins; ctx->exact.n = n; ctx->exact.ndiffuse = ndiffuse; ctx->exact.nimages = nimages;
    return RVB_OK;
}

// Exact mode, step 2: bins [b0, b1) — every bin's impulses added in impulse order ON TOP of what the histogram holds (rayverb.cpp:67-74).
static int exact_fold(rvb_ctx * ctx, uint64_t b0, uint64_t b1, float * hist)
{
    const AttenuationModel & m = ctx->model;
    const rvb_ctx::ExactState & e = ctx->exact;
    if (!e.valid) return fail(ctx, RVB_ERR_STATE, "rvb_ir_exact_fold: rvb_ir_exact_prepare first (and nothing that reuses the sort buffers in between)");
    if (b1 > e.nbins) b1 = e.nbins;
    if (e.hrtf_combined) {
        const uint64_t nkeys = 2 * (e.nbins + 1);
        rvb_launch_ordered_sum_hrtf(m, ir_diffuse(ctx), e.ndiffuse, ctx->images.as<rvb_impulse>(), ctx->vals_b.as<uint32_t>(),
                                    ctx->bin_starts.as<uint32_t>(), ctx->bin_starts.as<uint32_t>() + nkeys, e.nbins, hist, ctx->stream, b0, b1);
    } else {
        rvb_launch_ordered_sum(m, 0, m.nchannels, ir_diffuse(ctx), e.ndiffuse, ctx->images.as<rvb_impulse>(), e.nimages, ctx->vals_b.as<uint32_t>(),
                               ctx->bin_starts.as<uint32_t>(), ctx->bin_starts.as<uint32_t>() + e.nbins, e.n, e.nbins, hist, ctx->stream, b0, b1);
    }
    RVB_HIP(ctx, hipGetLastError());
    return RVB_OK;
}

// Bins [b0, b1) of every row of the [rows][nbins] histogram leave for pinned host memory on the export stream, behind what the context's
// stream holds now (rvb_copy_to_pinned_host_async for a bin range: one strided copy).
static int export_bin_range(rvb_ctx * ctx, float * pinned_dst, const float * hist, uint64_t rows, uint64_t nbins, uint64_t b0, uint64_t b1)
{
    if (b1 <= b0) return RVB_OK;
    RVB_HIP(ctx, hipEventRecord(ctx->export_ready, ctx->stream));
    RVB_HIP(ctx, hipStreamWaitEvent(ctx->export_stream, ctx->export_ready, 0));
    if (b0 == 0 && b1 == nbins) {
        RVB_HIP(ctx, hipMemcpyAsync(pinned_dst, hist, rows * nbins * sizeof(float), hipMemcpyDeviceToHost, ctx->export_stream));
    } else {
        static const bool by_rows = getenv("RVB_EXPORT_ROWS") && getenv("RVB_EXPORT_ROWS")[0] == '1';      // measurement: one copy per [channel][band] row
        if (by_rows) {
            for (uint64_t r = 0; r < rows; ++r)
                RVB_HIP(ctx, hipMemcpyAsync(pinned_dst + r * nbins + b0, hist + r * nbins + b0, (b1 - b0) * sizeof(float), hipMemcpyDeviceToHost, ctx->export_stream));
        } else {
            RVB_HIP(ctx, hipMemcpy2DAsync(pinned_dst + b0, nbins * sizeof(float), hist + b0, nbins * sizeof(float), (b1 - b0) * sizeof(float), rows,
                                          hipMemcpyDeviceToHost, ctx->export_stream));
        }
    }
    return RVB_OK;
}

// rvb_ir_accumulate, and — with pinned_dst — the histogram's way to the host: in exact mode with the speaker model the last kernel of the
// stage (ordered_sum_kernel: a lane pair per bin) runs bin range by bin range and every range's copy is enqueued behind it, so the link
// is busy while the later ranges are still being folded; the other forms copy the finished histogram in one piece.
static int ir_accumulate_impl(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins, int mode, void * d_histogram,
                              float * pinned_dst, uint32_t slices)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->ir_configured) return fail(ctx, RVB_ERR_STATE, "rvb_ir_accumulate: rvb_ir_configure_* first");
    if (!d_histogram || nbins == 0) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_accumulate: null histogram or no bins");
    RVB_BIND(ctx);
    if (slices == 0) slices = 1;
    bool exported = false;
    const AttenuationModel & m = ctx->model;
    const uint64_t ndiffuse = (ctx->which & RVB_IR_DIFFUSE) ? ctx->nrays * ctx->nreflections : 0;
    const uint64_t nimages = (ctx->which & RVB_IR_IMAGES) ? ctx->nimages : 0;
    float * hist = reinterpret_cast<float *>(d_histogram);
    ctx->reset_timings();
    if (mode == RVB_IR_FAST) {
        const size_t acc_bytes = (size_t) nbins * m.nchannels * 8 * sizeof(float);
        RVB_HIP(ctx, ctx->acc.ensure(acc_bytes));
        RVB_HIP(ctx, hipMemsetAsync(ctx->acc.p, 0, acc_bytes, ctx->stream));
        ctx->begin_timing("histogram_fast_kernel");
        rvb_launch_histogram_fast(m, ir_diffuse(ctx), ndiffuse, predelay, sample_rate, nbins, ctx->acc.as<float>(), ctx->stream);
        rvb_launch_histogram_fast(m, ctx->images.as<rvb_impulse>(), nimages, predelay, sample_rate, nbins, ctx->acc.as<float>(), ctx->stream);
        ctx->end_timing();
        ctx->begin_timing("histogram_transpose_kernel");
        rvb_launch_histogram_transpose(ctx->acc.as<float>(), hist, m.nchannels, nbins, ctx->stream);
        ctx->end_timing();
    } else if (mode == RVB_IR_EXACT) {
        const char * split_env = getenv("RVB_HRTF_SPLIT_EARS");     // measurement / test switch (read per call): one list per ear, as in round 2
        const bool split_ears = m.hrtf && split_env && split_env[0] == '1';
        ctx->begin_timing("exact_mode");
        if (!split_ears) {
            // one sorted list (speaker channels share it; the two HRTF ears are keyed into one list of 2 n entries), then the fold —
            // bin range by bin range when the histogram leaves for the host as it becomes final
            int rc = exact_prepare(ctx, predelay, sample_rate, nbins);
            if (rc != RVB_OK) return rc;
            const uint32_t parts = pinned_dst ? slices : 1u;
            const uint64_t per = ((nbins + parts - 1) / parts + 15) & ~15ull;      // whole 64-byte segments
            for (uint64_t b0 = 0; b0 < nbins; b0 += per) {
                const uint64_t b1 = std::min(nbins, b0 + per);
                rc = exact_fold(ctx, b0, b1, hist);
                if (rc == RVB_OK && pinned_dst && parts > 1) rc = export_bin_range(ctx, pinned_dst, hist, (uint64_t) m.nchannels * 8, nbins, b0, b1);
                if (rc != RVB_OK) return rc;
            }
            exported = pinned_dst && parts > 1;
        } else {
            const uint64_t n = ndiffuse + nimages;
            if (n >= (1ull << 31)) return fail(ctx, RVB_ERR_CAPACITY, "rvb_ir_accumulate: too many impulses for exact mode");
            if (nbins >= 0x7FFFFFF0ull) return fail(ctx, RVB_ERR_CAPACITY, "rvb_ir_accumulate: too many bins for exact mode");
            ctx->flat_host = nullptr;
            int rc = ensure_sort_buffers(ctx, n);
            if (rc != RVB_OK) return rc;
            const int bits = key_bits_for(nbins);
            const uint32_t sentinel = (uint32_t) nbins;
            RVB_HIP(ctx, ctx->bin_starts.ensure(nbins * 8));          // starts, then ends
            for (uint32_t ch = 0; ch < m.nchannels; ++ch) {           // a list per ear
                rvb_launch_bin_keys(m, ch, ir_diffuse(ctx), ndiffuse, 0, predelay, sample_rate, sentinel,
                                    ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), ctx->stream);
                rvb_launch_bin_keys(m, ch, ctx->images.as<rvb_impulse>(), nimages, ndiffuse, predelay, sample_rate, sentinel,
                                    ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), ctx->stream);
                rvb_sort_pairs(ctx->sort_temp.p, ctx->sort_temp.cap, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(),
                               ctx->vals_a.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), n, bits, ctx->stream);
                RVB_HIP(ctx, hipMemsetAsync(ctx->bin_starts.p, 0xFF, nbins * 4, ctx->stream));
                rvb_launch_bin_starts(ctx->keys_b.as<uint32_t>(), n, nbins, ctx->bin_starts.as<uint32_t>(), ctx->bin_starts.as<uint32_t>() + nbins, ctx->stream);
                rvb_launch_ordered_sum(m, ch, 1u, ir_diffuse(ctx), ndiffuse, ctx->images.as<rvb_impulse>(), nimages,
                                       ctx->vals_b.as<uint32_t>(), ctx->bin_starts.as<uint32_t>(), ctx->bin_starts.as<uint32_t>() + nbins, n, nbins, hist, ctx->stream);
            }
        }
        ctx->end_timing();
    } else {
        return fail(ctx, RVB_ERR_INVALID, "rvb_ir_accumulate: unknown mode");
    }
    RVB_HIP(ctx, hipGetLastError());
    if (pinned_dst && !exported) return export_bin_range(ctx, pinned_dst, hist, (uint64_t) m.nchannels * 8, nbins, 0, nbins);
    return RVB_OK;
}

int rvb_ir_accumulate(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins, int mode, void * d_histogram)
{
    return ir_accumulate_impl(ctx, predelay, sample_rate, nbins, mode, d_histogram, nullptr, 1);
}

int rvb_ir_accumulate_export(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins, int mode, void * d_histogram,
                             void * pinned_dst, uint32_t slices)
{
    if (ctx && !pinned_dst) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_accumulate_export: null destination");
    static const uint32_t env_slices = getenv("RVB_EXPORT_SLICES") ? (uint32_t) atoi(getenv("RVB_EXPORT_SLICES")) : 0;      // measurements
    // Default: ONE piece.  Measured at workload C2 (profiles/r04_export_slices_n1.txt): 1 / 2 / 4 / 8 bin ranges leave one impulse response
    // on the host after 6.98 / 7.04 / 7.02 / 7.00 ms (5.88 ms to HBM: the 54 MB need 1.1 ms of the link whenever they start, and the fold they
    // could overlap is 0.27 ms split into launches that cost what the overlap gains) and the pipeline at 4.82 / 4.81 / 4.80 / 4.96 ms per IR.
    return ir_accumulate_impl(ctx, predelay, sample_rate, nbins, mode, d_histogram, static_cast<float *>(pinned_dst), env_slices ? env_slices : (slices ? slices : 1u));
}

int rvb_ir_exact_prepare(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!ctx->ir_configured) return fail(ctx, RVB_ERR_STATE, "rvb_ir_exact_prepare: rvb_ir_configure_* first");
    if (nbins == 0) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_exact_prepare: no bins");
    RVB_BIND(ctx);
    ctx->reset_timings();
    ctx->begin_timing("exact_prepare");
    const int rc = exact_prepare(ctx, predelay, sample_rate, nbins);
    ctx->end_timing();
    return rc;
}

int rvb_ir_exact_fold(rvb_ctx * ctx, uint64_t nbins, uint64_t bin_begin, uint64_t bin_end, void * d_histogram)
{
    if (!ctx) return RVB_ERR_INVALID;
    if (!d_histogram) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_exact_fold: null histogram");
    if (!ctx->exact.valid || ctx->exact.nbins != nbins) return fail(ctx, RVB_ERR_STATE, "rvb_ir_exact_fold: rvb_ir_exact_prepare with this nbins first");
    if (bin_begin > bin_end) return fail(ctx, RVB_ERR_INVALID, "rvb_ir_exact_fold: bin range");
    RVB_BIND(ctx);
    return exact_fold(ctx, bin_begin, bin_end, static_cast<float *>(d_histogram));
}

int rvb_ir_download(rvb_ctx * ctx, int trim_predelay, float sample_rate, int mode,
                    float * out, uint64_t capacity_bins, uint64_t * nbins)
{
    if (!ctx || !nbins) return RVB_ERR_INVALID;
    float lo = 0.0f, hi = 0.0f;
    int rc = rvb_ir_time_range(ctx, &lo, &hi);
    if (rc != RVB_OK) return rc;
    const float predelay = trim_predelay ? lo : 0.0f;
    const uint64_t bins = bins_for(hi, predelay, sample_rate);
    *nbins = bins;
    if (!out)
        return RVB_OK;
    if (capacity_bins < bins)
        return fail(ctx, RVB_ERR_CAPACITY, "rvb_ir_download: capacity_bins too small");
    const size_t bytes = (size_t) bins * ctx->model.nchannels * 8 * sizeof(float);
    RVB_HIP(ctx, ctx->hist.ensure(bytes));
    RVB_HIP(ctx, hipMemsetAsync(ctx->hist.p, 0, bytes, ctx->stream));
    rc = rvb_ir_accumulate(ctx, predelay, sample_rate, bins, mode, ctx->hist.p);
    if (rc != RVB_OK) return rc;
    RVB_HIP(ctx, hipMemcpyAsync(out, ctx->hist.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RVB_OK;
}

int rvb_last_timings(rvb_ctx * ctx, char * names, uint64_t names_capacity, float * ms, uint64_t ms_capacity, uint64_t * count)
{
    if (!ctx || !count) return RVB_ERR_INVALID;
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::string joined;
    uint64_t n = 0;
    for (const Timing & t : ctx->timings) {
        float v = 0.0f;
        RVB_HIP(ctx, hipEventElapsedTime(&v, t.start, t.stop));
        if (ms && n < ms_capacity) ms[n] = v;
        if (!joined.empty()) joined += ';';
        joined += t.name;
        ++n;
    }
    *count = n;
    if (names && names_capacity) {
        std::strncpy(names, joined.c_str(), names_capacity - 1);
        names[names_capacity - 1] = 0;
    }
    return RVB_OK;
}

int rvb_debug_stamps(rvb_ctx * ctx, uint64_t * out, uint64_t capacity)
{
    if (!ctx || !out) return RVB_ERR_INVALID;
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_debug_stamps: nothing traced");
    RVB_BIND(ctx);
    RVB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long v[32];
    RVB_HIP(ctx, hipMemcpy(v, ctx->stamps.p, sizeof(v), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < capacity && i < 32; ++i) out[i] = v[i];
    return RVB_OK;
}

int rvb_executed_bounces(rvb_ctx * ctx, uint64_t * bounces)
{
    if (!ctx || !bounces) return RVB_ERR_INVALID;
    if (!ctx->traced) return fail(ctx, RVB_ERR_STATE, "rvb_executed_bounces: nothing traced");
    RVB_BIND(ctx);
    int rc = fetch_small(ctx);
    if (rc != RVB_OK) return rc;
    unsigned long long v = 0;
    std::memcpy(&v, ctx->small_host + kSmallExecuted, sizeof(v));
    *bounces = v;
    return RVB_OK;
}

}  // extern "C"
