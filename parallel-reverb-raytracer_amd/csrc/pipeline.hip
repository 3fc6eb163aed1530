// pipeline.hip — impulse responses back to back behind the C-ABI (rvb_pipeline_* of include/rvb_capi.h): the schedule that
// bench.py times, for a C / C++ caller.  What it replaces for a batch caller is the reference's cmd/main.cpp:241-298 in a loop — one
// Raytracer::raytrace, one Attenuator::attenuate per channel, fixPredelay, flattenImpulses per impulse response, each stage blocking.
//
// The trace of an impulse response is bound by vector issue, its record grouping and binning by memory, its image-source merge and its
// configuration by the host.  So several contexts of one GPU (the caller's: same scene, same rays on each) take turns:
//   * jobs are traced in GROUPS of `group` contexts — one path-kernel launch for the group (rvb_trace_group): more waves per SIMD for
//     the latency-bound bounce chains;
//   * the traces of the group after next are enqueued before the current group is finished, so a path kernel is (nearly) always
//     resident and the other stages of the previous group run beside it;
//   * the binning stages of ALL impulse responses of a group are enqueued — each behind its own trace's image-source candidates —
//     before the host waits for any of them;
//   * every finished histogram leaves for a pinned host buffer of the pipeline's ring on the context's export stream, bin range by
//     bin range (rvb_ir_accumulate_export); rvb_pipeline_next returns when the oldest submitted job's histogram has landed.
// Single host thread, nothing but the public C-ABI underneath (plus HIP for the histogram buffers and their zero fill).
#include "../../include/rvb_capi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

namespace {

struct Job {
    uint64_t id = 0;
    float mic[3] = {0, 0, 0}, source[3] = {0, 0, 0}, facing[3] = {0, 0, 0}, up[3] = {0, 0, 0};
    bool begun = false, staged = false;
    uint64_t nbins = 0, nimages = 0;
    float predelay = 0.0f, max_time = 0.0f;
    float * host = nullptr;
};

struct Slot {                                 // per context
    rvb_ctx * ctx = nullptr;
    bool has_table = false;                   // the pipeline's HRTF table is on this context's device (uploaded with its first HRTF job)
    float * hist = nullptr;                   // device [nchannels][8][nbins]
    size_t hist_cap = 0;
    hipEvent_t zeroed = nullptr;
};

struct HostBuffer { float * p = nullptr; size_t cap = 0; };

}  // namespace

struct rvb_pipeline {
    std::vector<Slot> slots;
    uint64_t group = 1;
    int device = 0;
    hipStream_t fill_stream = nullptr;        // zero fills of the histograms (the contexts' streams wait for them by an event)
    std::string error;
    // model + binning configuration
    bool configured = false, hrtf = false;
    std::vector<rvb_speaker> speakers;
    std::vector<float> table;                 // [2][360*180*8]
    float facing[3] = {0, 0, 1}, up[3] = {0, 1, 0};
    int which = RVB_IR_ALL, remove_direct = 0, trim_predelay = 1, mode = RVB_IR_EXACT;
    float sample_rate = 44100.0f;
    uint64_t nreflections = 0;
    float air[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // jobs: [returned, submitted); jobs.front() is the oldest not yet returned
    std::deque<Job> jobs;
    uint64_t submitted = 0, returned = 0, begun_upto = 0;
    std::vector<HostBuffer> ring;             // pinned result buffers, job id % ring.size()
    std::vector<rvb_image_candidate> candidates;
    std::vector<rvb_impulse> images;
};

namespace {

int pfail(rvb_pipeline * p, int code, const std::string & what)
{
    if (p) p->error = what;
    return code;
}
int cfail(rvb_pipeline * p, int code, rvb_ctx * ctx, const char * where)
{
    return pfail(p, code, std::string(where) + ": " + rvb_last_error(ctx));
}
#define PHIP(p, call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return pfail(p, RVB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

Job & job_at(rvb_pipeline * p, uint64_t id) { return p->jobs[(size_t) (id - p->returned)]; }

// the traces of jobs [first, last): one launch for the group where the contexts allow it (rvb_trace_group decides)
int begin_jobs(rvb_pipeline * p, uint64_t first, uint64_t last)
{
    const uint64_t n = p->slots.size(), count = last - first;
    rvb_ctx * ctxs[RVB_PIPELINE_MAX_GROUP];
    float mics[3 * RVB_PIPELINE_MAX_GROUP], sources[3 * RVB_PIPELINE_MAX_GROUP];
    for (uint64_t k = 0; k < count; ++k) {
        const Job & j = job_at(p, first + k);
        ctxs[k] = p->slots[(size_t) ((first + k) % n)].ctx;
        std::memcpy(mics + 3 * k, j.mic, sizeof(j.mic));
        std::memcpy(sources + 3 * k, j.source, sizeof(j.source));
    }
    int rc;
    if (count > 1) rc = rvb_trace_group(ctxs, count, mics, sources, p->nreflections, p->air, nullptr);
    else rc = rvb_trace(ctxs[0], mics, sources, p->nreflections, p->air, 0);
    if (rc != RVB_OK) return cfail(p, rc, ctxs[0], "rvb_pipeline: trace");
    for (uint64_t k = 0; k < count; ++k) job_at(p, first + k).begun = true;
    return RVB_OK;
}

// begins jobs in submission order, group by group, up to (not including) job `limit`
int begin_upto(rvb_pipeline * p, uint64_t limit)
{
    limit = std::min(limit, p->submitted);
    // job i runs on context i % contexts: it may go out once the job that used the context before has handed it over — returned, or at
    // least staged (its binning is enqueued; the new trace follows it in stream order and touches none of its buffers)
    const uint64_t n = p->slots.size();
    for (uint64_t id = p->begun_upto; id < limit; ++id)
        if (id >= n && id - n >= p->returned && !job_at(p, id - n).staged) { limit = id; break; }
    while (p->begun_upto < limit) {
        const uint64_t group_end = (p->begun_upto / p->group + 1) * p->group;
        const uint64_t last = std::min(group_end, limit);
        const int rc = begin_jobs(p, p->begun_upto, last);
        if (rc != RVB_OK) return rc;
        p->begun_upto = last;
    }
    return RVB_OK;
}

// Staging of a traced job in two phases, so that a group's jobs overlap their device work: (1) image-source merge and configuration — the
// only host wait is the one for the trace's small result block (image-source candidates, time range of the speaker model) — and the HRTF
// model's time-range pass ENQUEUED; (2) the time range read, the histogram sized and zeroed, binning + export enqueued.
int stage_configure(rvb_pipeline * p, Job & j)
{
    Slot & s = p->slots[(size_t) (j.id % p->slots.size())];
    rvb_ctx * ctx = s.ctx;
    uint64_t ncand = 0, nimages = 0;
    int rc = rvb_get_image_candidates(ctx, nullptr, 0, &ncand);
    if (rc != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: candidates");
    p->candidates.resize(ncand);
    if (ncand && (rc = rvb_get_image_candidates(ctx, p->candidates.data(), ncand, &ncand)) != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: candidates");
    p->images.clear();
    if (p->which & RVB_IR_IMAGES) {
        rvb_impulse direct;
        if ((rc = rvb_get_direct(ctx, &direct)) != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: direct path");
        if ((rc = rvb_merge_images(p->candidates.data(), ncand, &direct, p->remove_direct, nullptr, 0, &nimages)) != RVB_OK) return pfail(p, rc, "rvb_pipeline: rvb_merge_images");
        p->images.resize(nimages);
        if (nimages && (rc = rvb_merge_images(p->candidates.data(), ncand, &direct, p->remove_direct, p->images.data(), nimages, &nimages)) != RVB_OK)
            return pfail(p, rc, "rvb_pipeline: rvb_merge_images");
    }
    if (p->hrtf) {
        // (the table goes up with a context's first job only: rvb_ir_configure_hrtf keeps it for table == NULL)
        rc = rvb_ir_configure_hrtf(ctx, j.mic, s.has_table ? nullptr : p->table.data(), j.facing, j.up, p->which, p->images.data(), nimages);
        s.has_table = rc == RVB_OK;
    }
    else rc = rvb_ir_configure_speakers(ctx, j.mic, p->speakers.data(), p->speakers.size(), p->which, p->images.data(), nimages);
    if (rc != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: configure");
    j.nimages = nimages;
    if ((rc = rvb_ir_time_range_begin(ctx)) != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: time range");
    return RVB_OK;
}

int stage_bin(rvb_pipeline * p, Job & j)
{
    Slot & s = p->slots[(size_t) (j.id % p->slots.size())];
    rvb_ctx * ctx = s.ctx;
    float lo = 0.0f, hi = 0.0f;
    int rc = rvb_ir_time_range(ctx, &lo, &hi);
    if (rc != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: time range");
    j.predelay = p->trim_predelay ? lo : 0.0f;
    j.max_time = hi;
    j.nbins = rvb_ir_bins(hi, j.predelay, p->sample_rate);
    const uint64_t nch = p->hrtf ? 2 : p->speakers.size();
    const size_t bytes = (size_t) j.nbins * nch * 8 * sizeof(float);
    PHIP(p, hipSetDevice(p->device));
    if (bytes > s.hist_cap) {
        // (the previous histogram of this context left for the host before its result was handed out: nothing reads it any more)
        if (s.hist) { PHIP(p, hipFree(s.hist)); s.hist = nullptr; s.hist_cap = 0; }
        PHIP(p, hipMalloc(reinterpret_cast<void **>(&s.hist), bytes + bytes / 8));
        s.hist_cap = bytes + bytes / 8;
    }
    HostBuffer & hb = p->ring[(size_t) (j.id % p->ring.size())];
    if (bytes > hb.cap) {
        if (hb.p) { PHIP(p, hipHostFree(hb.p)); hb.p = nullptr; hb.cap = 0; }
        PHIP(p, hipHostMalloc(reinterpret_cast<void **>(&hb.p), bytes + bytes / 8, hipHostMallocDefault));
        hb.cap = bytes + bytes / 8;
    }
    j.host = hb.p;
    PHIP(p, hipMemsetAsync(s.hist, 0, bytes, p->fill_stream));
    PHIP(p, hipEventRecord(s.zeroed, p->fill_stream));
    if ((rc = rvb_wait_for_event(ctx, s.zeroed)) != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: wait for the zero fill");
    if ((rc = rvb_ir_accumulate_export(ctx, j.predelay, p->sample_rate, j.nbins, p->mode, s.hist, j.host, 0)) != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline: binning");
    j.staged = true;
    return RVB_OK;
}

int configure_common(rvb_pipeline * p, int which, int remove_direct, int trim_predelay, float sample_rate, int mode, uint64_t nreflections,
                     const float air[8])
{
    if (!p->jobs.empty()) return pfail(p, RVB_ERR_STATE, "rvb_pipeline_configure: jobs are pending");
    if (which < 1 || which > 3) return pfail(p, RVB_ERR_INVALID, "rvb_pipeline_configure: which must be 1..3");
    if (mode != RVB_IR_FAST && mode != RVB_IR_EXACT) return pfail(p, RVB_ERR_INVALID, "rvb_pipeline_configure: unknown mode");
    if (!air || nreflections == 0 || !(sample_rate > 0.0f)) return pfail(p, RVB_ERR_INVALID, "rvb_pipeline_configure: reflections, sample rate, air coefficients");
    p->which = which; p->remove_direct = remove_direct; p->trim_predelay = trim_predelay; p->sample_rate = sample_rate; p->mode = mode;
    p->nreflections = nreflections;
    std::memcpy(p->air, air, sizeof(p->air));
    p->configured = true;
    return RVB_OK;
}

}  // namespace

extern "C" {

int rvb_pipeline_create(rvb_pipeline ** out, rvb_ctx ** ctxs, uint64_t count, uint64_t group)
{
    if (!out || !ctxs || count == 0 || count > 64) return RVB_ERR_INVALID;
    *out = nullptr;
    for (uint64_t i = 0; i < count; ++i) {
        if (!ctxs[i]) return RVB_ERR_INVALID;
        for (uint64_t k = 0; k < i; ++k)
            if (ctxs[k] == ctxs[i]) return RVB_ERR_INVALID;
    }
    rvb_pipeline * p = new rvb_pipeline();
    // groups of half the contexts (two groups in flight), as measured best at workload C2 (4 contexts in groups of 2); at most what one
    // launch takes
    uint64_t g = group ? group : std::max<uint64_t>(1, count / 2);
    g = std::min<uint64_t>(std::min<uint64_t>(g, count), RVB_PIPELINE_MAX_GROUP);
    p->group = g;
    // (the contexts are the caller's: scene and rays set, all on one device)
    int device = 0;
    for (uint64_t i = 0; i < count; ++i) {
        int d = 0;
        if (rvb_device_index(ctxs[i], &d) != RVB_OK || (i && d != device)) { delete p; return RVB_ERR_INVALID; }
        device = d;
    }
    p->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete p; return RVB_ERR_HIP; }
    if (hipStreamCreateWithFlags(&p->fill_stream, hipStreamNonBlocking) != hipSuccess) { delete p; return RVB_ERR_HIP; }
    for (uint64_t i = 0; i < count; ++i) {
        Slot s;
        s.ctx = ctxs[i];
        if (hipEventCreateWithFlags(&s.zeroed, hipEventDisableTiming) != hipSuccess) { rvb_pipeline_destroy(p); return RVB_ERR_HIP; }
        p->slots.push_back(s);
        (void) rvb_set_concurrent_traces(ctxs[i], (uint32_t) g);     // the traces of a group run side by side: the path kernel is sized for them
    }
    p->ring.resize((size_t) (2 * count));
    *out = p;
    return RVB_OK;
}

void rvb_pipeline_destroy(rvb_pipeline * p)
{
    if (!p) return;
    (void) hipSetDevice(p->device);
    for (Slot & s : p->slots) {
        if (s.ctx) { (void) rvb_synchronize(s.ctx); (void) rvb_synchronize_exports(s.ctx); (void) rvb_set_concurrent_traces(s.ctx, 1); }
        if (s.hist) (void) hipFree(s.hist);
        if (s.zeroed) (void) hipEventDestroy(s.zeroed);
    }
    for (HostBuffer & h : p->ring)
        if (h.p) (void) hipHostFree(h.p);
    if (p->fill_stream) { (void) hipStreamSynchronize(p->fill_stream); (void) hipStreamDestroy(p->fill_stream); }
    delete p;
}

const char * rvb_pipeline_last_error(const rvb_pipeline * p) { return p ? p->error.c_str() : ""; }

int rvb_pipeline_configure_speakers(rvb_pipeline * p, const rvb_speaker * speakers, uint64_t nspeakers, int which, int remove_direct,
                                    int trim_predelay, float sample_rate, int mode, uint64_t nreflections, const float air_coefficient[8])
{
    if (!p) return RVB_ERR_INVALID;
    if (!speakers || nspeakers == 0 || nspeakers > 8) return pfail(p, RVB_ERR_INVALID, "rvb_pipeline_configure_speakers: 1..8 speakers required");
    const int rc = configure_common(p, which, remove_direct, trim_predelay, sample_rate, mode, nreflections, air_coefficient);
    if (rc != RVB_OK) return rc;
    p->hrtf = false;
    p->speakers.assign(speakers, speakers + nspeakers);
    return RVB_OK;
}

int rvb_pipeline_configure_hrtf(rvb_pipeline * p, const float * table, const float facing[3], const float up[3], int which, int remove_direct,
                                int trim_predelay, float sample_rate, int mode, uint64_t nreflections, const float air_coefficient[8])
{
    if (!p) return RVB_ERR_INVALID;
    if (!table || !facing || !up) return pfail(p, RVB_ERR_INVALID, "rvb_pipeline_configure_hrtf: null argument");
    const int rc = configure_common(p, which, remove_direct, trim_predelay, sample_rate, mode, nreflections, air_coefficient);
    if (rc != RVB_OK) return rc;
    p->hrtf = true;
    p->table.assign(table, table + (size_t) 2 * 360 * 180 * 8);
    for (Slot & s : p->slots) s.has_table = false;
    std::memcpy(p->facing, facing, sizeof(p->facing));
    std::memcpy(p->up, up, sizeof(p->up));
    return RVB_OK;
}

int rvb_pipeline_submit_oriented(rvb_pipeline * p, const float mic[3], const float source[3], const float facing[3], const float up[3])
{
    if (!p) return RVB_ERR_INVALID;
    if (!p->configured) return pfail(p, RVB_ERR_STATE, "rvb_pipeline_submit: rvb_pipeline_configure_* first");
    if (!mic || !source) return pfail(p, RVB_ERR_INVALID, "rvb_pipeline_submit: null argument");
    if (p->submitted - p->returned >= 4 * p->slots.size()) return pfail(p, RVB_ERR_CAPACITY, "rvb_pipeline_submit: take results first (4 x contexts jobs are pending)");
    Job j;
    j.id = p->submitted;
    std::memcpy(j.mic, mic, sizeof(j.mic));
    std::memcpy(j.source, source, sizeof(j.source));
    std::memcpy(j.facing, facing ? facing : p->facing, sizeof(j.facing));
    std::memcpy(j.up, up ? up : p->up, sizeof(j.up));
    p->jobs.push_back(j);
    ++p->submitted;
    // keep the device busy without waiting for the caller's next call: the traces of COMPLETE groups go out as soon as contexts are
    // free for them (job i runs on context i % contexts; the contexts of jobs that have not been returned yet are taken); an
    // incomplete last group is traced when rvb_pipeline_next gets to it
    return begin_upto(p, std::min(p->returned / p->group * p->group + p->slots.size(), p->submitted / p->group * p->group));
}

int rvb_pipeline_submit(rvb_pipeline * p, const float mic[3], const float source[3])
{
    return rvb_pipeline_submit_oriented(p, mic, source, nullptr, nullptr);
}

uint64_t rvb_pipeline_pending(const rvb_pipeline * p) { return p ? p->submitted - p->returned : 0; }

int rvb_pipeline_next(rvb_pipeline * p, rvb_pipeline_result * out)
{
    if (!p || !out) return RVB_ERR_INVALID;
    if (p->jobs.empty()) return pfail(p, RVB_ERR_STATE, "rvb_pipeline_next: nothing is pending");
    const uint64_t n = p->slots.size();
    Job & j = p->jobs.front();
    const uint64_t group_first = j.id / p->group * p->group;
    // the traces of the groups after this one, as far as contexts are free: every context holds one job
    int rc = begin_upto(p, group_first + n);
    if (rc != RVB_OK) return rc;
    if (!j.staged) {
        // the binning stages of all (begun) jobs of this group, before the host waits for any of them
        const uint64_t last = std::min(group_first + p->group, p->begun_upto);
        for (uint64_t id = j.id; id < last; ++id)
            if ((rc = stage_configure(p, job_at(p, id))) != RVB_OK) return rc;
        for (uint64_t id = j.id; id < last; ++id)
            if ((rc = stage_bin(p, job_at(p, id))) != RVB_OK) return rc;
        // ... then for their binning (not for their histograms' way to the host): with it done, the group's contexts take the traces of
        // the group after next — enqueued before this call waits for the link, so the copy runs beside them
        // (measured: enqueuing those traces in stream order behind the binning, BEFORE this wait — RVB_PIPELINE_EARLY_TRACES=1 — costs 4.45 -> 5.6 ms per
        // IR at workload C2: a third group's path kernel then competes with this group's binning for the SIMDs)
        static const bool early = getenv("RVB_PIPELINE_EARLY_TRACES") && getenv("RVB_PIPELINE_EARLY_TRACES")[0] == '1';
        if (early && (rc = begin_upto(p, std::min(group_first + p->group + n, p->submitted / p->group * p->group))) != RVB_OK) return rc;
        for (uint64_t id = j.id; id < last; ++id) {
            rvb_ctx * c = p->slots[(size_t) (id % n)].ctx;
            if ((rc = rvb_synchronize(c)) != RVB_OK) return cfail(p, rc, c, "rvb_pipeline_next: wait");
        }
        // (and enqueuing them LATER costs as well: a host delay of 100 / 300 / 600 / 1000 us here: 4.45 -> 4.53 / 4.57 / 4.69 / 4.96 ms per IR)
        if ((rc = begin_upto(p, std::min(group_first + p->group + n, p->submitted / p->group * p->group))) != RVB_OK) return rc;
    }
    rvb_ctx * ctx = p->slots[(size_t) (j.id % n)].ctx;
    if ((rc = rvb_synchronize_exports(ctx)) != RVB_OK) return cfail(p, rc, ctx, "rvb_pipeline_next: wait");
    out->job = j.id;
    out->histogram = j.host;
    out->nchannels = p->hrtf ? 2 : p->speakers.size();
    out->nbins = j.nbins;
    out->predelay = j.predelay;
    out->max_time = j.max_time;
    out->nimages = j.nimages;
    p->jobs.pop_front();
    ++p->returned;
    return rc;
}

}  // extern "C"
