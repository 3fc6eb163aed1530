// multi.hip — several GPUs of one node behind the C-ABI (rvb_multi_* of include/rvb_capi.h): one rvb_ctx and one host
// thread per device, contiguous ray shards, image-source candidates merged with the reference's lowest-ray-wins rule
// (rayverb.cpp:654-676), histograms combined on the devices.  The reference is single-device (rayverb.cpp:163, :176-177);
// this is the fan-out SURVEY.md §8(b) / §8(e) ask for, without Python or torch in the way.
//
// Histogram modes over D devices:
//   RVB_IR_EXACT  a CHAIN: device 0 folds its impulses into a zeroed histogram, hands it to device 1 (peer copy), which
//                 continues the same left-to-right float sum with ITS impulses (rvb_ir_accumulate adds on top of what the
//                 histogram holds), and so on; the merged image sources go last.  Shards are consecutive ray ranges, so the
//                 chain is the reference's serial order over all rays: bit-identical to one context (and to flattenImpulses).
//                 The traces — 85 % of the time — still run side by side, and so does every device's keying and sorting
//                 (rvb_ir_exact_prepare: it does not depend on the incoming histogram).  Only the fold is a chain, and it is
//                 SYSTOLIC (round 4): the histogram travels in B bin-range blocks, device g folds block k while device g + 1
//                 folds block k - 1 — (D + B - 1) / B folds and hops instead of D; events, no host waits between devices.
//   RVB_IR_FAST   every device bins its shard at once (float atomics), then ONE sum over devices: RCCL ncclAllReduce over
//                 xGMI, loaded from librccl.so at run time; devices that RCCL cannot put in one communicator (the same GPU
//                 listed twice, as the single-GPU tests do) are summed with peer copies and an add kernel instead.
#include "../../include/rvb_capi.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---- the four RCCL entry points used, bound at run time (prototypes: /opt/rocm/include/rccl/rccl.h:236, :260, :339, :611, :919) ----
typedef struct ncclComm * ncclComm_t;
struct Rccl {
    void * lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && GetErrorString && AllReduce && GroupStart && GroupEnd; }
};
const int kNcclFloat = 7, kNcclSum = 0;       // rccl.h:466, :448

Rccl & rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char * names[] = {getenv("RVB_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char * n : names) {
            if (!n) continue;
            x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (x.lib) {
            x.CommInitAll = reinterpret_cast<int (*)(ncclComm_t *, int, const int *)>(dlsym(x.lib, "ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<int (*)(ncclComm_t)>(dlsym(x.lib, "ncclCommDestroy"));
            x.GetErrorString = reinterpret_cast<const char * (*)(int)>(dlsym(x.lib, "ncclGetErrorString"));
            x.AllReduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t)>(dlsym(x.lib, "ncclAllReduce"));
            x.GroupStart = reinterpret_cast<int (*)()>(dlsym(x.lib, "ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<int (*)()>(dlsym(x.lib, "ncclGroupEnd"));
        }
        return x;
    }();
    return r;
}

__global__ __launch_bounds__(256) void add_kernel(float * __restrict__ acc, const float * __restrict__ x, uint64_t n)
{
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        acc[i] += x[i];
}

struct Shard {
    rvb_ctx * ctx = nullptr;
    int device = 0;
    uint64_t first = 0, count = 0;            // ray range
    hipStream_t stream = nullptr;             // histogram traffic of this device
    float * hist = nullptr;
    size_t hist_cap = 0;
    float * peer = nullptr;                   // landing buffer for another device's histogram (fallback sum)
    size_t peer_cap = 0;
    std::vector<hipEvent_t> arrived, folded;  // exact chain: block k of the histogram has landed on this device / has been folded here
    int rc = RVB_OK;
    std::string error;
};

}  // namespace

struct rvb_multi {
    std::vector<Shard> shards;
    std::string error;
    uint64_t nrays = 0, nreflections = 0;
    bool traced = false;
    std::vector<ncclComm_t> comms;            // one per shard when RCCL serves this device list
    bool rccl_tried = false, rccl_used_last = false;
    unsigned flags = 0;
    uint32_t chain_blocks = 8;                // bin-range blocks of the exact chain (rvb_multi_set_chain_blocks)
    int peer_links = 0;                       // directed device pairs with peer access enabled (rvb_multi_create)
    float mic[3] = {0, 0, 0};
    std::vector<rvb_impulse> images;          // merged image sources of the last rvb_multi_ir_* call
};

namespace {

int mfail(rvb_multi * m, int code, const std::string & what)
{
    if (m) m->error = what;
    return code;
}

// f(shard) on one host thread per device; the first failure is reported
template <class F>
int for_each_shard(rvb_multi * m, F f)
{
    std::vector<std::thread> pool;
    for (Shard & s : m->shards)
        pool.emplace_back([&s, &f] {
            s.rc = f(s);
            if (s.rc != RVB_OK) s.error = rvb_last_error(s.ctx);
        });
    for (std::thread & t : pool) t.join();
    for (Shard & s : m->shards)
        if (s.rc != RVB_OK) return mfail(m, s.rc, "device " + std::to_string(s.device) + ": " + s.error);
    return RVB_OK;
}

hipError_t ensure(int device, float *& p, size_t & cap, size_t bytes)
{
    if (bytes <= cap) return hipSuccess;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    if (p) { (void) hipFree(p); p = nullptr; cap = 0; }
    e = hipMalloc(reinterpret_cast<void **>(&p), bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
}

#define MHIP(m, call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return mfail(m, RVB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

bool distinct_devices(const rvb_multi * m)
{
    for (size_t i = 0; i < m->shards.size(); ++i)
        for (size_t j = i + 1; j < m->shards.size(); ++j)
            if (m->shards[i].device == m->shards[j].device) return false;
    return true;
}

// one communicator over the device list, created on first use
bool ensure_rccl(rvb_multi * m)
{
    if (m->rccl_tried) return !m->comms.empty();
    m->rccl_tried = true;
    static const bool off = getenv("RVB_MULTI_RCCL") && getenv("RVB_MULTI_RCCL")[0] == '0';
    if (off || !rccl().ok() || !distinct_devices(m)) return false;
    std::vector<int> devs;
    for (const Shard & s : m->shards) devs.push_back(s.device);
    std::vector<ncclComm_t> comms(devs.size(), nullptr);
    if (rccl().CommInitAll(comms.data(), (int) devs.size(), devs.data()) != 0) return false;
    m->comms.swap(comms);
    return true;
}

}  // namespace

extern "C" {

int rvb_multi_create(rvb_multi ** out, const int * devices, int ndevices, unsigned flags)
{
    if (!out || ndevices <= 0 || ndevices > 64) return RVB_ERR_INVALID;
    *out = nullptr;
    rvb_multi * m = new rvb_multi();
    m->flags = flags;
    for (int i = 0; i < ndevices; ++i) {
        Shard s;
        s.device = devices ? devices[i] : i;
        const int rc = rvb_create(&s.ctx, s.device, 0);
        if (rc != RVB_OK) {
            for (Shard & made : m->shards) rvb_destroy(made.ctx);
            delete m;
            return rc;                         // rvb_last_error(NULL) holds the text
        }
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
            rvb_destroy(s.ctx);
            for (Shard & made : m->shards) rvb_destroy(made.ctx);
            delete m;
            return RVB_ERR_HIP;
        }
        m->shards.push_back(s);
    }
    // peer access between every pair of distinct devices (the chain's hops, the fallback sum's gathers): without it the runtime stages a
    // peer copy through the host.  A refusal is not an error — the copies still work, staged.
    for (const Shard & a : m->shards)
        for (const Shard & b : m->shards) {
            if (a.device == b.device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a.device, b.device) != hipSuccess || !can) { (void) hipGetLastError(); continue; }
            if (hipSetDevice(a.device) != hipSuccess) { (void) hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ++m->peer_links;
            (void) hipGetLastError();
        }
    *out = m;
    return RVB_OK;
}

int rvb_multi_set_chain_blocks(rvb_multi * m, uint32_t blocks)
{
    if (!m) return RVB_ERR_INVALID;
    if (blocks == 0 || blocks > 64) return mfail(m, RVB_ERR_INVALID, "rvb_multi_set_chain_blocks: 1 .. 64");
    m->chain_blocks = blocks;
    return RVB_OK;
}

int rvb_multi_peer_links(const rvb_multi * m) { return m ? m->peer_links : 0; }

void rvb_multi_destroy(rvb_multi * m)
{
    if (!m) return;
    for (ncclComm_t c : m->comms)
        if (c) (void) rccl().CommDestroy(c);
    for (Shard & s : m->shards) {
        (void) hipSetDevice(s.device);
        if (s.stream) { (void) hipStreamSynchronize(s.stream); (void) hipStreamDestroy(s.stream); }
        for (hipEvent_t e : s.arrived) (void) hipEventDestroy(e);
        for (hipEvent_t e : s.folded) (void) hipEventDestroy(e);
        if (s.hist) (void) hipFree(s.hist);
        if (s.peer) (void) hipFree(s.peer);
        rvb_destroy(s.ctx);
    }
    delete m;
}

const char * rvb_multi_last_error(const rvb_multi * m) { return m ? m->error.c_str() : rvb_last_error(nullptr); }

int rvb_multi_devices(const rvb_multi * m) { return m ? (int) m->shards.size() : 0; }

int rvb_multi_context(rvb_multi * m, int index, rvb_ctx ** ctx, uint64_t * first_ray, uint64_t * nrays)
{
    if (!m || index < 0 || index >= (int) m->shards.size()) return RVB_ERR_INVALID;
    if (ctx) *ctx = m->shards[(size_t) index].ctx;
    if (first_ray) *first_ray = m->shards[(size_t) index].first;
    if (nrays) *nrays = m->shards[(size_t) index].count;
    return RVB_OK;
}

int rvb_multi_used_rccl(const rvb_multi * m) { return m && m->rccl_used_last ? 1 : 0; }

int rvb_multi_set_scene(rvb_multi * m, const rvb_triangle * triangles, uint64_t ntriangles, const rvb_float3 * vertices, uint64_t nvertices,
                        const rvb_surface * surfaces, uint64_t nsurfaces)
{
    if (!m) return RVB_ERR_INVALID;
    m->traced = false;
    // replicated (MBs); every device builds the same tree from the same input
    return for_each_shard(m, [&](Shard & s) { return rvb_set_scene(s.ctx, triangles, ntriangles, vertices, nvertices, surfaces, nsurfaces); });
}

int rvb_multi_set_directions(rvb_multi * m, const rvb_float3 * directions, uint64_t nrays)
{
    if (!m) return RVB_ERR_INVALID;
    if (nrays && !directions) return mfail(m, RVB_ERR_INVALID, "rvb_multi_set_directions: null directions");
    const uint64_t d = m->shards.size(), base = nrays / d, extra = nrays % d;
    for (uint64_t g = 0; g < d; ++g) {         // contiguous ranges that differ by at most one ray (distributed.shard_range)
        m->shards[g].first = g * base + std::min(g, extra);
        m->shards[g].count = base + (g < extra ? 1 : 0);
    }
    m->nrays = nrays;
    m->traced = false;
    return for_each_shard(m, [&](Shard & s) { return rvb_set_directions(s.ctx, directions + s.first, s.count); });
}

int rvb_multi_trace(rvb_multi * m, const float mic[3], const float source[3], uint64_t nreflections, const float air_coefficient[8])
{
    if (!m) return RVB_ERR_INVALID;
    if (!mic || !source || !air_coefficient) return mfail(m, RVB_ERR_INVALID, "rvb_multi_trace: null argument");
    for (int i = 0; i < 3; ++i) m->mic[i] = mic[i];
    m->nreflections = nreflections;
    const int rc = for_each_shard(m, [&](Shard & s) {
        int r = rvb_trace(s.ctx, mic, source, nreflections, air_coefficient, s.first);       // ray numbers stay global
        return r != RVB_OK ? r : rvb_synchronize(s.ctx);
    });
    m->traced = rc == RVB_OK;
    return rc;
}

int rvb_multi_get_diffuse(rvb_multi * m, rvb_impulse * out)
{
    if (!m) return RVB_ERR_INVALID;
    if (!m->traced) return mfail(m, RVB_ERR_STATE, "rvb_multi_get_diffuse: nothing traced");
    if (m->nrays * m->nreflections && !out) return mfail(m, RVB_ERR_INVALID, "rvb_multi_get_diffuse: null output");
    // every device writes its slice of the ray-major array: D links at once
    return for_each_shard(m, [&](Shard & s) { return s.count ? rvb_get_diffuse(s.ctx, out + s.first * m->nreflections) : (int) RVB_OK; });
}

static int merged_images(rvb_multi * m, int remove_direct, std::vector<rvb_impulse> & images)
{
    std::vector<rvb_image_candidate> all;
    for (Shard & s : m->shards) {
        uint64_t n = 0;
        int rc = rvb_get_image_candidates(s.ctx, nullptr, 0, &n);
        if (rc != RVB_OK) return mfail(m, rc, rvb_last_error(s.ctx));
        const size_t at = all.size();
        all.resize(at + n);
        if (n && (rc = rvb_get_image_candidates(s.ctx, all.data() + at, n, &n)) != RVB_OK) return mfail(m, rc, rvb_last_error(s.ctx));
    }
    rvb_impulse direct;
    std::memset(&direct, 0, sizeof(direct));
    int rc = rvb_get_direct(m->shards[0].ctx, &direct);       // the same on every device
    if (rc != RVB_OK) return mfail(m, rc, rvb_last_error(m->shards[0].ctx));
    const rvb_impulse * direct_ptr = m->nrays ? &direct : nullptr;
    uint64_t count = 0;
    if ((rc = rvb_merge_images(all.data(), all.size(), direct_ptr, remove_direct, nullptr, 0, &count)) != RVB_OK) return mfail(m, rc, "rvb_merge_images");
    images.resize(count);
    if ((rc = rvb_merge_images(all.data(), all.size(), direct_ptr, remove_direct, images.data(), count, &count)) != RVB_OK) return mfail(m, rc, "rvb_merge_images");
    return RVB_OK;
}

int rvb_multi_get_images(rvb_multi * m, int remove_direct, rvb_impulse * out, uint64_t capacity, uint64_t * count)
{
    if (!m || !count) return RVB_ERR_INVALID;
    if (!m->traced) return mfail(m, RVB_ERR_STATE, "rvb_multi_get_images: nothing traced");
    std::vector<rvb_impulse> images;
    const int rc = merged_images(m, remove_direct, images);
    if (rc != RVB_OK) return rc;
    *count = images.size();
    if (!out) return RVB_OK;
    if (capacity < images.size()) return mfail(m, RVB_ERR_CAPACITY, "rvb_multi_get_images: capacity too small");
    if (!images.empty()) std::memcpy(out, images.data(), images.size() * sizeof(rvb_impulse));
    return RVB_OK;
}

// model: speakers != NULL -> speaker channels, else HRTF (table, facing, up)
static int multi_ir(rvb_multi * m, const float mic[3], const rvb_speaker * speakers, uint64_t nspeakers,
                    const float * table, const float * facing, const float * up,
                    int which, int remove_direct, int trim_predelay, float sample_rate, int mode,
                    float * out, uint64_t capacity_bins, uint64_t * nbins)
{
    if (!m || !nbins) return RVB_ERR_INVALID;
    if (!m->traced) return mfail(m, RVB_ERR_STATE, "rvb_multi_ir: nothing traced");
    if (which < 1 || which > 3) return mfail(m, RVB_ERR_INVALID, "rvb_multi_ir: which must be 1..3");
    const uint32_t nch = speakers ? (uint32_t) nspeakers : 2u;
    auto configure = [&](rvb_ctx * ctx, int w, const rvb_impulse * images, uint64_t nimages) {
        return speakers ? rvb_ir_configure_speakers(ctx, mic, speakers, nspeakers, w, images, nimages)
                        : rvb_ir_configure_hrtf(ctx, mic, table, facing, up, w, images, nimages);
    };
    // 1. merged image sources (host, a few dozen records); time range of every shard and of the images
    m->images.clear();
    if (which & RVB_IR_IMAGES) {
        const int rc = merged_images(m, remove_direct, m->images);
        if (rc != RVB_OK) return rc;
    }
    std::vector<float> lo(m->shards.size() + 1, 0.0f), hi(m->shards.size() + 1, 0.0f);
    if (which & RVB_IR_DIFFUSE) {
        const int rc = for_each_shard(m, [&](Shard & s) {
            const size_t g = (size_t) (&s - m->shards.data());
            int r = configure(s.ctx, RVB_IR_DIFFUSE, nullptr, 0);
            return r != RVB_OK ? r : rvb_ir_time_range(s.ctx, &lo[g], &hi[g]);
        });
        if (rc != RVB_OK) return rc;
    }
    Shard & last = m->shards.back();
    if (!m->images.empty()) {
        int rc = configure(last.ctx, RVB_IR_IMAGES, m->images.data(), m->images.size());
        if (rc == RVB_OK) rc = rvb_ir_time_range(last.ctx, &lo.back(), &hi.back());
        if (rc != RVB_OK) return mfail(m, rc, rvb_last_error(last.ctx));
    }
    float min_nonzero = 0.0f, max_time = 0.0f;                 // findPredelay / MAX_SAMPLE inputs over all shards
    for (size_t i = 0; i < lo.size(); ++i) {
        if (lo[i] > 0.0f && (min_nonzero == 0.0f || lo[i] < min_nonzero)) min_nonzero = lo[i];
        max_time = std::max(max_time, hi[i]);
    }
    const float predelay = trim_predelay ? min_nonzero : 0.0f;
    const uint64_t bins = rvb_ir_bins(max_time, predelay, sample_rate);
    *nbins = bins;
    if (!out) return RVB_OK;
    if (capacity_bins < bins) return mfail(m, RVB_ERR_CAPACITY, "rvb_multi_ir: capacity_bins too small");
    const size_t count = (size_t) bins * nch * 8, bytes = count * sizeof(float);
    for (Shard & s : m->shards) MHIP(m, ensure(s.device, s.hist, s.hist_cap, bytes));
    m->rccl_used_last = false;
    Shard * result = nullptr;
    if (mode == RVB_IR_EXACT) {
        // 2a. the chain, systolic: every device keys and sorts its impulses at once; the histogram then travels in B bin-range blocks
        // (whole 64-byte segments of every [channel][band] row): device g folds block k behind its arrival from device g - 1, while
        // device g + 1 folds block k - 1.  Everything is enqueued from this thread in dependency order (an event is recorded before the
        // wait that names it is enqueued); the host waits once, at the end.
        const uint64_t rows = (uint64_t) nch * 8;
        const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(m->chain_blocks, (bins + 15) / 16));
        const uint64_t per = ((bins + blocks - 1) / blocks + 15) & ~15ull;
        const uint64_t nblocks = (bins + per - 1) / per;
        for (Shard & s : m->shards) {
            MHIP(m, hipSetDevice(s.device));
            while (s.arrived.size() < nblocks) { hipEvent_t e; MHIP(m, hipEventCreateWithFlags(&e, hipEventDisableTiming)); s.arrived.push_back(e); }
            while (s.folded.size() < nblocks) { hipEvent_t e; MHIP(m, hipEventCreateWithFlags(&e, hipEventDisableTiming)); s.folded.push_back(e); }
        }
        const int rc_prep = for_each_shard(m, [&](Shard & s) {
            if (!(which & RVB_IR_DIFFUSE) || !s.count) return (int) RVB_OK;
            int r = configure(s.ctx, RVB_IR_DIFFUSE, nullptr, 0);
            return r != RVB_OK ? r : rvb_ir_exact_prepare(s.ctx, predelay, sample_rate, bins);
        });
        if (rc_prep != RVB_OK) return rc_prep;
        for (size_t g = 0; g < m->shards.size(); ++g) {
            Shard & s = m->shards[g];
            const bool folds = (which & RVB_IR_DIFFUSE) && s.count;
            MHIP(m, hipSetDevice(s.device));
            if (g == 0) MHIP(m, hipMemsetAsync(s.hist, 0, bytes, s.stream));
            for (uint64_t k = 0; k < nblocks; ++k) {
                const uint64_t b0 = k * per, b1 = std::min<uint64_t>(bins, b0 + per);
                if (g > 0) {
                    Shard & prev = m->shards[g - 1];
                    MHIP(m, hipStreamWaitEvent(s.stream, prev.folded[k], 0));
                    if (prev.device == s.device) {
                        MHIP(m, hipMemcpy2DAsync(s.hist + b0, bins * sizeof(float), prev.hist + b0, bins * sizeof(float), (b1 - b0) * sizeof(float), rows,
                                                 hipMemcpyDeviceToDevice, s.stream));
                    } else {
                        for (uint64_t r = 0; r < rows; ++r)     // (a row's piece of the block is contiguous: one peer copy each)
                            MHIP(m, hipMemcpyPeerAsync(s.hist + r * bins + b0, s.device, prev.hist + r * bins + b0, prev.device, (b1 - b0) * sizeof(float), s.stream));
                    }
                }
                MHIP(m, hipEventRecord(s.arrived[k], s.stream));
                if (folds) {
                    int rc = rvb_wait_for_event(s.ctx, s.arrived[k]);
                    if (rc == RVB_OK) rc = rvb_ir_exact_fold(s.ctx, bins, b0, b1, s.hist);
                    if (rc == RVB_OK) rc = rvb_record_event(s.ctx, s.folded[k]);
                    if (rc != RVB_OK) return mfail(m, rc, rvb_last_error(s.ctx));
                } else {
                    MHIP(m, hipEventRecord(s.folded[k], s.stream));          // nothing to add here: the block passes through
                }
            }
        }
        // the last device's folds are the end of the chain (a device without impulses only passed blocks on its own stream)
        MHIP(m, hipSetDevice(last.device));
        MHIP(m, hipStreamSynchronize(last.stream));
        { const int rc = rvb_synchronize(last.ctx); if (rc != RVB_OK) return mfail(m, rc, rvb_last_error(last.ctx)); }
        result = &last;
    } else if (mode == RVB_IR_FAST) {
        // 2b. all shards at once, then one sum over the devices
        int rc = for_each_shard(m, [&](Shard & s) {
            if (hipSetDevice(s.device) != hipSuccess || hipMemsetAsync(s.hist, 0, bytes, s.stream) != hipSuccess ||
                hipStreamSynchronize(s.stream) != hipSuccess)
                return (int) RVB_ERR_HIP;
            if (!(which & RVB_IR_DIFFUSE) || !s.count) return (int) RVB_OK;
            int r = configure(s.ctx, RVB_IR_DIFFUSE, nullptr, 0);
            if (r == RVB_OK) r = rvb_ir_accumulate(s.ctx, predelay, sample_rate, bins, RVB_IR_FAST, s.hist);
            return r != RVB_OK ? r : rvb_synchronize(s.ctx);
        });
        if (rc != RVB_OK) return rc;
        const bool force = (m->flags & RVB_MULTI_REHEARSE_RCCL) != 0;
        if ((m->shards.size() > 1 || force) && ensure_rccl(m)) {
            // RCCL over xGMI: [channels][8][nbins] floats, in place on every device (rccl.h:611)
            int e = rccl().GroupStart();
            for (size_t g = 0; g < m->shards.size() && e == 0; ++g) {
                MHIP(m, hipSetDevice(m->shards[g].device));
                e = rccl().AllReduce(m->shards[g].hist, m->shards[g].hist, count, kNcclFloat, kNcclSum, m->comms[g], m->shards[g].stream);
            }
            const int e2 = rccl().GroupEnd();
            if (e != 0 || e2 != 0) return mfail(m, RVB_ERR_HIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(e ? e : e2));
            for (Shard & s : m->shards) { MHIP(m, hipSetDevice(s.device)); MHIP(m, hipStreamSynchronize(s.stream)); }
            m->rccl_used_last = true;
        } else if (m->shards.size() > 1) {
            // no communicator for this device list: gather on shard 0 with peer copies, add there
            Shard & root = m->shards[0];
            MHIP(m, ensure(root.device, root.peer, root.peer_cap, bytes));
            MHIP(m, hipSetDevice(root.device));
            for (size_t g = 1; g < m->shards.size(); ++g) {
                MHIP(m, hipMemcpyPeerAsync(root.peer, root.device, m->shards[g].hist, m->shards[g].device, bytes, root.stream));
                hipLaunchKernelGGL(add_kernel, dim3(4096), dim3(256), 0, root.stream, root.hist, root.peer, (uint64_t) count);
                MHIP(m, hipGetLastError());
            }
            MHIP(m, hipStreamSynchronize(root.stream));
        }
        result = &m->shards[0];
    } else {
        return mfail(m, RVB_ERR_INVALID, "rvb_multi_ir: unknown mode");
    }
    // 3. the merged image sources go last (reference order: diffuse, then images — rayverb.cpp:708-714)
    if (!m->images.empty()) {
        int rc = configure(result->ctx, RVB_IR_IMAGES, m->images.data(), m->images.size());
        if (rc == RVB_OK) rc = rvb_ir_accumulate(result->ctx, predelay, sample_rate, bins, mode, result->hist);
        if (rc == RVB_OK) rc = rvb_synchronize(result->ctx);
        if (rc != RVB_OK) return mfail(m, rc, rvb_last_error(result->ctx));
    }
    return rvb_copy_to_host(result->ctx, out, result->hist, bytes) == RVB_OK ? (int) RVB_OK : mfail(m, RVB_ERR_HIP, rvb_last_error(result->ctx));
}

int rvb_multi_ir_speakers(rvb_multi * m, const float mic[3], const rvb_speaker * speakers, uint64_t nspeakers, int which, int remove_direct,
                          int trim_predelay, float sample_rate, int mode, float * out, uint64_t capacity_bins, uint64_t * nbins)
{
    if (!m) return RVB_ERR_INVALID;
    if (!mic || !speakers || nspeakers == 0 || nspeakers > 8) return mfail(m, RVB_ERR_INVALID, "rvb_multi_ir_speakers: 1..8 speakers required");
    return multi_ir(m, mic, speakers, nspeakers, nullptr, nullptr, nullptr, which, remove_direct, trim_predelay, sample_rate, mode, out, capacity_bins, nbins);
}

int rvb_multi_ir_hrtf(rvb_multi * m, const float mic[3], const float * table, const float facing[3], const float up[3], int which, int remove_direct,
                      int trim_predelay, float sample_rate, int mode, float * out, uint64_t capacity_bins, uint64_t * nbins)
{
    if (!m) return RVB_ERR_INVALID;
    if (!mic || !table || !facing || !up) return mfail(m, RVB_ERR_INVALID, "rvb_multi_ir_hrtf: null argument");
    return multi_ir(m, mic, nullptr, 0, table, facing, up, which, remove_direct, trim_predelay, sample_rate, mode, out, capacity_bins, nbins);
}

}  // extern "C"
