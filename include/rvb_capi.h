/* rvb_capi.h — C-ABI of the MI355X-native acoustic ray tracer (librvb_hip.so).
 *
 * This is the drop-in boundary for the reference's per-ray hot path.  The reference has no
 * FFI of its own for this path: its callers (cmd/main.cpp:241-298 and the gtest fixtures)
 * use the C++ classes of rayverb/rayverb.h directly, which in turn drive three OpenCL
 * kernels.  Each entry point below names the reference interface it replaces; the C++ mirror
 * of those classes (include/rayverb/rayverb.h, built on nothing but this header) is what a
 * maintainer links instead of the OpenCL-backed library — see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns RVB_OK (0) or an RVB_ERR_* code; rvb_last_error() gives the text
 *     (the reference throws cl::Error / std::runtime_error instead, rayverb.cpp:151-192);
 *   - all pointers are caller-owned; "host" pointers are ordinary memory, "device" pointers are
 *     HBM addresses on the context's GPU (e.g. torch.Tensor.data_ptr());
 *   - PODs are layout-identical to reference rayverb/clstructs.h (sizes in SURVEY.md §8(a));
 *     this header demands no particular alignment of host buffers;
 *   - a context is bound to one GPU and one HIP stream and is not thread-safe (the reference's
 *     objects are not either: one in-order queue per KernelLoader, rayverb.cpp:191);
 *   - there is no CPU fallback: without a usable gfx950 device rvb_create() fails.
 */
#ifndef RVB_CAPI_H
#define RVB_CAPI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVB_NUM_IMAGE_SOURCE 10   /* reference rayverb/clstructs.h:4 */
#define RVB_NUM_BANDS 8           /* VolumeType = cl_float8, reference rayverb/clstructs.h:13 */

enum {
    RVB_OK = 0,
    RVB_ERR_INVALID = 1,     /* bad argument */
    RVB_ERR_NO_DEVICE = 2,   /* no usable gfx950 GPU / HIP runtime failure at start-up */
    RVB_ERR_HIP = 3,         /* a HIP call failed */
    RVB_ERR_STATE = 4,       /* call order (e.g. trace before set_scene) */
    RVB_ERR_CAPACITY = 5     /* caller buffer too small / scene exceeds a built-in limit */
};

/* reference rayverb/clstructs.h:17-24 (Triangle), :27-32 (Surface), :36-42 (Impulse),
 * :44-49 (AttenuatedImpulse), :53-58 (Speaker); cl_float3 is a 16-byte float4. */
typedef struct { uint64_t surface, v0, v1, v2; } rvb_triangle;                          /* 32 B */
typedef struct { float s[4]; } rvb_float3;                                              /* 16 B */
typedef struct { float specular[8]; float diffuse[8]; } rvb_surface;                    /* 64 B */
typedef struct { float volume[8]; float position[4]; float time; float pad_[3]; } rvb_impulse;     /* 64 B */
typedef struct { float volume[8]; float time; float pad_[7]; } rvb_attenuated_impulse;  /* 64 B */
typedef struct { float direction[4]; float coefficient; float pad_[3]; } rvb_speaker;   /* 32 B */

/* One valid image-source contribution of one ray (what the reference keeps per work-item in
 * image_source[i*10+slot] / image_source_index[i*10+slot], kernel.cpp:258-264), compacted. */
typedef struct {
    uint64_t ray;            /* global ray index (ray_offset of the trace call added) */
    uint32_t slot;           /* 1..9 (slot 0, the direct path, is reported separately) */
    uint32_t index;          /* triangle index + 1 (kernel.cpp:453) */
    rvb_impulse impulse;
} rvb_image_candidate;       /* 80 B */

typedef struct rvb_ctx rvb_ctx;

/* ---- life cycle: replaces ContextProvider / KernelLoader (rayverb.cpp:151-192) ------------- */
int rvb_create(rvb_ctx ** out, int device, unsigned flags);
void rvb_destroy(rvb_ctx * ctx);
const char * rvb_last_error(const rvb_ctx * ctx);       /* ctx may be NULL after a failed rvb_create */
int rvb_synchronize(rvb_ctx * ctx);                     /* wait for the context's stream */
/* Work submitted to the context after this call starts only after `hip_event` (a hipEvent_t recorded on another stream,
 * e.g. the one a caller-owned buffer was zeroed on) has completed.  No host synchronisation. */
int rvb_wait_for_event(rvb_ctx * ctx, void * hip_event);
/* Records `hip_event` (a hipEvent_t of the caller's) behind the work submitted to the context so far: the counterpart of rvb_wait_for_event
 * for a caller that hands a buffer the context has written to a stream of its own (e.g. a histogram block to the next device). */
int rvb_record_event(rvb_ctx * ctx, void * hip_event);
/* Name ("gfx950"), compute-unit count and HBM bytes of the bound device; its HIP device index. */
int rvb_device_info(rvb_ctx * ctx, char * arch, uint64_t arch_capacity, int * compute_units, uint64_t * hbm_bytes);
int rvb_device_index(rvb_ctx * ctx, int * device);

/* ---- scene: replaces the geometry half of Raytracer::Raytracer (rayverb.cpp:242-293) --------
 * Copies triangles / vertices / surfaces, builds the BVH on the host and uploads everything.
 * Triangle and surface indices are validated (the reference never calls its own
 * SceneData::valid(), rayverb.cpp:463-502; out-of-range indices are rejected here). */
int rvb_set_scene(rvb_ctx * ctx,
                  const rvb_triangle * triangles, uint64_t ntriangles,
                  const rvb_float3 * vertices, uint64_t nvertices,
                  const rvb_surface * surfaces, uint64_t nsurfaces);
/* Gives `ctx` the scene `from` holds — the SAME device buffers, no second build, no second copy: the contexts of one GPU that trace
 * side by side in one room (rvb_pipeline_*, rvb_trace_group; the reference builds one Raytracer per scene, rayverb.cpp:242-293) then
 * keep one hierarchy in HBM and in the L2s instead of one each.  The buffers live as long as any context holds them; a later
 * rvb_set_scene on either context gives THAT context a scene of its own and leaves the other's untouched.
 * RVB_ERR_STATE: `from` has no scene; RVB_ERR_INVALID: the contexts are on different devices. */
int rvb_share_scene(rvb_ctx * ctx, rvb_ctx * from);
/* BVH statistics of the current scene (nodes, leaf triangles kept, tree depth). */
int rvb_scene_info(rvb_ctx * ctx, uint64_t * nodes, uint64_t * kept_triangles, uint32_t * depth);

/* ---- ray directions: replaces the per-group cl::copy of rayverb.cpp:593-598 ---------------- */
/* Directions are unit vectors (reference getRandomDirections, helpers.cpp:63-81); RVB_ERR_INVALID for a direction that is not
 * finite or whose length is outside [0.5, 2] (the range the pruning margins of the acceleration structure are derived for).
 * The device variant borrows the caller's buffer and does not inspect it: the same contract is the caller's to keep. */
int rvb_set_directions(rvb_ctx * ctx, const rvb_float3 * directions, uint64_t nrays);          /* host */
int rvb_set_directions_device(rvb_ctx * ctx, const void * d_directions, uint64_t nrays);        /* device, borrowed */
/* A hint, never a change of results: how many traces of this size the caller keeps in flight on the device at a time (several
 * contexts whose streams run side by side; default 1).  The path kernel spends two lanes per ray instead of four when the rays in
 * flight fill the chip without the extra waves (csrc/trace_kernels.hip, rvb_path_lanes_for).  No reference counterpart: the
 * reference runs one 4096-ray group at a time (rayverb.cpp:586-591). */
int rvb_set_concurrent_traces(rvb_ctx * ctx, uint32_t traces);
/* Measurement / test hook, never a change of results: the path kernel of this context's traces with `lanes` lanes per ray — 4 (path_kernel),
 * 2 (path_pair_kernel), 1 (path_lane_kernel) — whatever the launch size; 0 (default) lets every launch choose (rvb_path_lanes_for).  The three
 * kernels write the same bytes (tests/test_gpu_parity.py runs every trace case with each). */
int rvb_set_path_lanes(rvb_ctx * ctx, uint32_t lanes);

/* ---- trace: replaces Raytracer::raytrace (rayverb.cpp:538-685) + kernel raytrace
 * (kernel.cpp:304-503).  Traces exactly nrays rays (all at once, no 4096-ray groups) for
 * nreflections bounces; results stay in HBM.  Asynchronous on the context's stream.
 * ray_offset is added to ray numbers reported by rvb_get_image_candidates (multi-GPU shards). */
int rvb_trace(rvb_ctx * ctx, const float mic[3], const float source[3],
              uint64_t nreflections, const float air_coefficient[8], uint64_t ray_offset);

/* Several (source, microphone) pairs of one scene in ONE launch — what a caller of the reference does with one
 * Raytracer::raytrace (rayverb.cpp:538-685) per pair, e.g. the 64 pairs of a hall.  Every pair is traced with the context's
 * nrays directions; ray r of pair p is global ray p * nrays + r: rvb_get_diffuse / rvb_diffuse_device return
 * [npairs][nrays][nreflections] impulses and rvb_get_image_candidates reports global ray numbers (pair = ray / nrays).
 * mics / sources are [npairs][3].  More rays per launch fill the GPU better (path tracing costs 3.8 ms per 100 k rays at
 * 100 k, 2.9 ms at 400 k); results are bit-identical to tracing the pairs one by one. */
int rvb_trace_pairs(rvb_ctx * ctx, const float * mics, const float * sources, uint64_t npairs,
                    uint64_t nreflections, const float air_coefficient[8], uint64_t ray_offset);

/* rvb_trace on several contexts of one device at once (at most 4): same results as calling rvb_trace(ctxs[i], mics + 3 i, sources + 3 i,
 * nreflections, air_coefficient, ray_offsets ? ray_offsets[i] : 0) one after the other, but the path kernels of the group are ONE launch
 * when the rays of the group fill the chip (their waves are then scheduled together instead of one launch after the other's).  Every
 * context keeps its own rays, buffers and stream; what follows the path kernel runs per context as in rvb_trace.  No reference
 * counterpart (the reference traces one 4096-ray group at a time, rayverb.cpp:586-591). */
int rvb_trace_group(rvb_ctx ** ctxs, uint64_t count, const float * mics, const float * sources, uint64_t nreflections,
                    const float air_coefficient[8], const uint64_t * ray_offsets);
/* Chooses the pair that rvb_get_direct and the rvb_ir_* calls below work on (pair 0 after a trace). */
int rvb_ir_select_pair(rvb_ctx * ctx, uint64_t pair);

/* ---- raw results: replace getRawDiffuse / getRawImages (rayverb.cpp:687-714) ---------------- */
int rvb_get_diffuse(rvb_ctx * ctx, rvb_impulse * out /* host [nrays*nreflections] */);
/* Device address of the same array (valid until the next rvb_trace / rvb_destroy). */
int rvb_diffuse_device(rvb_ctx * ctx, const void ** d_impulses, uint64_t * count);
/* Direct-path impulse (slot 0; all-zero when the source is not visible from the microphone). */
int rvb_get_direct(rvb_ctx * ctx, rvb_impulse * out);
/* Valid image-source contributions of this context's rays, sorted by (ray, slot). */
int rvb_get_image_candidates(rvb_ctx * ctx, rvb_image_candidate * out, uint64_t capacity, uint64_t * count);
/* Host-only merge = the de-dup map of rayverb.cpp:654-676 followed by getRawImages
 * (rayverb.cpp:692-706): first (lowest) ray wins per key, output in std::map key order.
 * Candidates of several contexts (GPU shards) may be concatenated before the call. */
int rvb_merge_images(const rvb_image_candidate * candidates, uint64_t ncandidates,
                     const rvb_impulse * direct, int remove_direct,
                     rvb_impulse * out, uint64_t capacity, uint64_t * count);

/* ---- attenuation, materialised: replace SpeakerAttenuator::attenuate (rayverb.cpp:856-892 +
 * kernel attenuate, kernel.cpp:505-535) and HrtfAttenuator::attenuate (rayverb.cpp:765-818 +
 * kernel hrtf, kernel.cpp:537-625) for ONE channel.  Host in, host out.  Impulses whose volume
 * is all-zero give {0, 0} (the reference leaves them uninitialised, quirk Q2). */
int rvb_attenuate_speaker(rvb_ctx * ctx, const float mic[3], const rvb_impulse * in, uint64_t n,
                          const rvb_speaker * speaker, rvb_attenuated_impulse * out);
/* The same kernel on device-resident buffers (e.g. rvb_diffuse_device): d_in / d_out are HBM arrays of n Impulse /
 * AttenuatedImpulse records, distinct.  Asynchronous on the context's stream; what SpeakerAttenuator::attenuate's
 * per-channel re-upload of the impulse array (rayverb.cpp:863-875) becomes when the trace results never leave HBM. */
int rvb_attenuate_speaker_device(rvb_ctx * ctx, const float mic[3], const void * d_in, uint64_t n,
                                 const rvb_speaker * speaker, void * d_out);
int rvb_attenuate_hrtf(rvb_ctx * ctx, const float mic[3], const rvb_impulse * in, uint64_t n,
                       const float * table /* [360*180*8] for this ear */,
                       const float facing[3], const float up[3], uint64_t channel,
                       rvb_attenuated_impulse * out);
/* ... and HrtfAttenuator::attenuate's kernel on device-resident buffers (table is host memory as above). */
int rvb_attenuate_hrtf_device(rvb_ctx * ctx, const float mic[3], const void * d_in, uint64_t n,
                              const float * table, const float facing[3], const float up[3], uint64_t channel, void * d_out);

/* ---- time binning, materialised: replaces flattenImpulses (rayverb.cpp:48-77) for one channel.
 * Bit-exact with the reference's serial summation order.  out is [8][*nbins].  Called with out == NULL it reports *nbins;
 * a following call with the same (in, n, sample_rate) finds the uploaded array and its keys still on the device, unless another
 * call on this context has used the sort buffers in between (then the fill uploads again).  PRECONDITION of that pair: the
 * caller does not change in[0 .. n) between the size query and the fill — the fill does not read the host array again. */
int rvb_flatten(rvb_ctx * ctx, const rvb_attenuated_impulse * in, uint64_t n, float sample_rate,
                float * out, uint64_t capacity_bins, uint64_t * nbins);

/* ---- fused, device-resident IR generation (attenuate + predelay + bin without materialising
 * AttenuatedImpulse): replaces the chain Attenuator::attenuate -> fixPredelay ->
 * flattenImpulses of cmd/main.cpp:280-298 + rayverb.h:49-97 + rayverb.cpp:28-77.
 *
 * 1. rvb_ir_configure_*   choose the attenuation model and which impulses take part;
 *                         extra image impulses (already merged) are uploaded here.
 * 2. rvb_ir_time_range    min non-zero / max attenuated time over all channels of this context
 *                         (the inputs of findPredelay and of MAX_SAMPLE); with several GPUs the
 *                         caller all-reduces (min, max) before step 3.
 * 3. rvb_ir_accumulate    adds this context's impulses into d_histogram [nchannels][8][nbins]
 *                         (device memory, caller-zeroed, float).  mode RVB_IR_FAST uses float
 *                         atomics (order-dependent in the last bits); RVB_IR_EXACT continues, bin by
 *                         bin, the reference's serial left-to-right float sum (rayverb.cpp:67-74) from
 *                         the values d_histogram holds: on a zeroed histogram that IS flattenImpulses
 *                         bit for bit, and contexts that hold consecutive ray shards reproduce the
 *                         single-context result when they accumulate into the same histogram in shard
 *                         order (diffuse shards first, the merged image sources last).
 */
enum { RVB_IR_DIFFUSE = 1, RVB_IR_IMAGES = 2, RVB_IR_ALL = 3 };   /* OutputMode, config.h:19-23 */
enum { RVB_IR_FAST = 0, RVB_IR_EXACT = 1 };

int rvb_ir_configure_speakers(rvb_ctx * ctx, const float mic[3], const rvb_speaker * speakers, uint64_t nspeakers,
                              int which, const rvb_impulse * images, uint64_t nimages);
/* table == NULL: the table of this context's previous rvb_ir_configure_hrtf call stays on the device (many listeners, one table: 4 MB
 * uploaded once instead of per impulse response); RVB_ERR_STATE if there is none. */
int rvb_ir_configure_hrtf(rvb_ctx * ctx, const float mic[3], const float * table /* [2][360*180*8] or NULL */,
                          const float facing[3], const float up[3],
                          int which, const rvb_impulse * images, uint64_t nimages);
int rvb_ir_time_range(rvb_ctx * ctx, float * min_nonzero_time, float * max_time);
/* Optional first half of rvb_ir_time_range: enqueues what the range needs on the device (the HRTF model's pass over the impulses; nothing
 * for speakers) without waiting; the rvb_ir_time_range that follows only waits and reads.  A caller that finishes several contexts'
 * impulse responses together calls this on all of them first (csrc/pipeline.hip). */
int rvb_ir_time_range_begin(rvb_ctx * ctx);
/* nbins for a given max time / predelay exactly as rayverb.cpp:57 computes MAX_SAMPLE. */
uint64_t rvb_ir_bins(float max_time, float predelay, float sample_rate);
int rvb_ir_accumulate(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins, int mode,
                      void * d_histogram);
/* rvb_ir_accumulate as above, and the finished [nchannels][8][nbins] histogram on its way to
 * pinned_dst (pinned host memory of the same size: rvb_host_alloc, hipHostMalloc, torch's pin_memory) on the context's export stream:
 * rvb_synchronize_exports waits for it, rvb_synchronize does not, the context's next trace does not either.  In exact mode with the
 * exact mode the last kernel of the stage can fold the histogram in `slices` bin ranges, every range leaving as soon as it is final
 * (the copy of all but the last range then runs beside the folding of the later ones); 0 = the default, ONE piece (= rvb_ir_accumulate +
 * rvb_copy_to_pinned_host_async): at workload C2 the ranges buy nothing — 54 MB need 1.1 ms of the link whenever they start, the fold
 * they could hide behind is 0.27 ms (profiles/r04_export_slices_n1.txt).  The float-atomic mode always copies the finished histogram
 * in one piece.  d_histogram must stay untouched until
 * the export has been waited for.  Reference counterpart: the blocking cl::copy of every result (rayverb.cpp:645-651, :678, :816). */
int rvb_ir_accumulate_export(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins, int mode,
                             void * d_histogram, void * pinned_dst, uint32_t slices);
/* RVB_IR_EXACT in two steps, for callers that hand the histogram on BLOCK BY BLOCK (the systolic chain over devices of csrc/multi.hip,
 * distributed.generate_ir(chain_exact=True)):
 *   rvb_ir_exact_prepare   everything that does not depend on what the histogram holds — bin keys, the radix sort, where each bin's run
 *                          starts and ends; every device of a chain does this as soon as its trace is done;
 *   rvb_ir_exact_fold      bins [bin_begin, bin_end): each bin's impulses added in impulse order on top of what d_histogram holds
 *                          (rayverb.cpp:67-74) — folding all bins once equals rvb_ir_accumulate(..., RVB_IR_EXACT, ...).
 * The prepared list lives in the context's sort buffers: a call that reuses them (rvb_ir_configure_*, rvb_trace, rvb_flatten*,
 * rvb_ir_accumulate) voids it and rvb_ir_exact_fold then fails with RVB_ERR_STATE.  Both are asynchronous on the context's stream. */
int rvb_ir_exact_prepare(rvb_ctx * ctx, float predelay, float sample_rate, uint64_t nbins);
int rvb_ir_exact_fold(rvb_ctx * ctx, uint64_t nbins, uint64_t bin_begin, uint64_t bin_end, void * d_histogram);
/* Convenience for one GPU: steps 1-3 done, histogram copied to host memory [nchannels][8][*nbins]. */
int rvb_ir_download(rvb_ctx * ctx, int trim_predelay, float sample_rate, int mode,
                    float * out, uint64_t capacity_bins, uint64_t * nbins);

/* ---- device-resident plumbing for the host mirror of the reference classes (host/rayverb_api.cpp) ------------------
 * The reference's API hands every stage's result over as a std::vector (rayverb.h:123-133, :290-293, rayverb.cpp:28-77);
 * its own implementation re-uploads those vectors stage by stage (rayverb.cpp:863-875).  A host that still holds the
 * device copy of a vector it produced can skip the re-upload: these calls give it the buffers and the kernels on them. */
int rvb_device_alloc(rvb_ctx * ctx, uint64_t bytes, void ** d_ptr);
int rvb_device_free(rvb_ctx * ctx, void * d_ptr);
/* Pageable host memory <-> HBM through pinned bounce buffers driven by several host threads (a plain hipMemcpy into fresh
 * pageable memory runs at a fraction of the link rate).  Synchronous; ordered after the work already on the context's stream. */
int rvb_copy_to_host(rvb_ctx * ctx, void * dst, const void * d_src, uint64_t bytes);
int rvb_copy_to_device(rvb_ctx * ctx, void * d_dst, const void * src, uint64_t bytes);
/* Pinned (page-locked, device-mapped) host memory, and a device -> pinned-host copy that is ASYNCHRONOUS: it starts when the work
 * enqueued so far on the context's stream has finished (e.g. the rvb_ir_accumulate that fills d_src) and runs on a stream of its
 * own, so the context's next trace does not wait for the link; rvb_synchronize_exports waits for it (rvb_synchronize does not).
 * d_src must stay untouched until then.  This is what carries a finished [nchannels][8][nbins] histogram to the host.
 * pinned_dst should be pinned memory (rvb_host_alloc, hipHostMalloc, torch's pin_memory): into pageable memory the runtime's
 * copy is staged and blocks the caller.  The reference's counterpart is the blocking cl::copy of every result buffer
 * (rayverb.cpp:645-651, :678, :816). */
int rvb_host_alloc(rvb_ctx * ctx, uint64_t bytes, void ** host_ptr);
int rvb_host_free(rvb_ctx * ctx, void * host_ptr);
int rvb_copy_to_pinned_host_async(rvb_ctx * ctx, void * pinned_dst, const void * d_src, uint64_t bytes);
int rvb_synchronize_exports(rvb_ctx * ctx);
/* fixPredelay (rayverb.h:76-90) on a device-resident AttenuatedImpulse array: time = time > seconds ? time - seconds : 0. */
int rvb_fix_predelay_device(rvb_ctx * ctx, void * d_attenuated, uint64_t n, float seconds);
/* rvb_flatten on a device-resident AttenuatedImpulse array (same result, no upload). */
int rvb_flatten_device(rvb_ctx * ctx, const void * d_attenuated, uint64_t n, float sample_rate,
                       float * out, uint64_t capacity_bins, uint64_t * nbins);

/* ---- several GPUs of one node (csrc/multi.hip) ---------------------------------------------------------------------------
 * The reference drives ONE device (rayverb.cpp:163, :176-177).  An rvb_multi owns one rvb_ctx and one host thread per listed
 * device: the scene is replicated, device g traces the contiguous ray range [g N / D, (g+1) N / D) of the directions handed to
 * rvb_multi_set_directions, image-source candidates are merged with the reference's lowest-ray-wins rule (rayverb.cpp:654-676)
 * and the per-band histograms are combined on the devices:
 *   RVB_IR_EXACT  the devices continue ONE left-to-right float sum in ray order (a chain of peer copies): bit-identical to a
 *                 single context and to the reference's flattenImpulses;
 *   RVB_IR_FAST   every device bins its shard at once, then one RCCL ncclAllReduce(sum) over xGMI (librccl.so is loaded at run
 *                 time; a device list RCCL cannot serve — the same GPU twice — is summed with peer copies instead).
 * devices == NULL means 0 .. ndevices-1.  Not thread-safe, like rvb_ctx. */
typedef struct rvb_multi rvb_multi;
enum { RVB_MULTI_REHEARSE_RCCL = 1 };     /* flags: run the RCCL all-reduce even for a single device (exercises the library binding) */
int rvb_multi_create(rvb_multi ** out, const int * devices, int ndevices, unsigned flags);
void rvb_multi_destroy(rvb_multi * m);
const char * rvb_multi_last_error(const rvb_multi * m);
int rvb_multi_devices(const rvb_multi * m);
/* The context of device slot `index` and the ray range it traced (e.g. to run independent (source, listener) pairs per device). */
int rvb_multi_context(rvb_multi * m, int index, rvb_ctx ** ctx, uint64_t * first_ray, uint64_t * nrays);
/* RVB_IR_EXACT over several devices is a SYSTOLIC chain: the [nchannels][8][nbins] histogram travels in `blocks` bin-range blocks
 * (default 8), device g folds block k while device g + 1 folds block k - 1 — (D + blocks - 1) / blocks folds and hops instead of D.
 * Any block count gives the same bytes (tests/test_gpu_multi.py forces 1, 3 and 8).  rvb_multi_peer_links: directed pairs of distinct
 * devices for which rvb_multi_create could enable peer access (0 on one GPU); without it the runtime stages the hops through the host. */
int rvb_multi_set_chain_blocks(rvb_multi * m, uint32_t blocks);
int rvb_multi_peer_links(const rvb_multi * m);
int rvb_multi_used_rccl(const rvb_multi * m);            /* 1 if the last RVB_IR_FAST histogram was summed by RCCL */
int rvb_multi_set_scene(rvb_multi * m, const rvb_triangle * triangles, uint64_t ntriangles, const rvb_float3 * vertices, uint64_t nvertices,
                        const rvb_surface * surfaces, uint64_t nsurfaces);
int rvb_multi_set_directions(rvb_multi * m, const rvb_float3 * directions, uint64_t nrays);
/* Raytracer::raytrace on all devices at once (blocking, like the reference's). */
int rvb_multi_trace(rvb_multi * m, const float mic[3], const float source[3], uint64_t nreflections, const float air_coefficient[8]);
/* getRawDiffuse / getRawImages over all shards: ray-major [nrays * nreflections]; merged image sources in std::map key order. */
int rvb_multi_get_diffuse(rvb_multi * m, rvb_impulse * out);
int rvb_multi_get_images(rvb_multi * m, int remove_direct, rvb_impulse * out, uint64_t capacity, uint64_t * count);
/* attenuate -> fixPredelay -> flattenImpulses over all shards; out (host) is [nchannels][8][*nbins]; out == NULL reports *nbins. */
int rvb_multi_ir_speakers(rvb_multi * m, const float mic[3], const rvb_speaker * speakers, uint64_t nspeakers, int which, int remove_direct,
                          int trim_predelay, float sample_rate, int mode, float * out, uint64_t capacity_bins, uint64_t * nbins);
int rvb_multi_ir_hrtf(rvb_multi * m, const float mic[3], const float * table /* [2][360*180*8] */, const float facing[3], const float up[3],
                      int which, int remove_direct, int trim_predelay, float sample_rate, int mode,
                      float * out, uint64_t capacity_bins, uint64_t * nbins);

/* ---- impulse responses back to back (csrc/pipeline.hip) --------------------------------------------------------------------
 * What a batch caller of the reference does with cmd/main.cpp:241-298 in a loop — raytrace, attenuate per channel, fixPredelay,
 * flattenImpulses per impulse response, every stage blocking — as a pipeline over several contexts of ONE GPU.  The contexts are the
 * caller's and stay the caller's: every one holds the same scene (rvb_set_scene) and the same rays (rvb_set_directions*) before the
 * pipeline is created, and is not used for anything else while jobs are pending.  Job i runs on context i % count.  Jobs are traced
 * in groups of `group` contexts with ONE path-kernel launch per group (rvb_trace_group; 0 = count / 2, at most
 * RVB_PIPELINE_MAX_GROUP), the traces of the group after next go out before the current group is finished, the binning stages of a
 * group are enqueued together, and every histogram leaves for pinned host memory on its context's export stream, bin range by bin
 * range (rvb_ir_accumulate_export).  Measured at workload C2 with 4 contexts: the rate bench.py reports.
 *   rvb_pipeline_configure_*   the attenuation model and the binning of all jobs that follow (no jobs may be pending)
 *   rvb_pipeline_submit        one impulse response: microphone and source (HRTF: the configured facing / up; _oriented: its own).
 *                              Never blocks; RVB_ERR_CAPACITY when 4 x count jobs are pending (take results first)
 *   rvb_pipeline_next          blocks until the OLDEST pending job's [nchannels][8][nbins] histogram is in host memory; the result's
 *                              `histogram` points into the pipeline's ring of pinned buffers and stays valid until `count` further
 *                              results have been taken (or the pipeline is destroyed)
 * Results are those of rvb_trace + rvb_merge_images + rvb_ir_configure_* + rvb_ir_download on one context, bit for bit in
 * RVB_IR_EXACT (tests/cpp/test_pipeline.cpp).  Not thread-safe. */
#define RVB_PIPELINE_MAX_GROUP 4
typedef struct rvb_pipeline rvb_pipeline;
typedef struct {
    uint64_t job;                 /* submission number: 0, 1, 2, ... */
    const float * histogram;      /* pinned host memory, [nchannels][8][nbins] */
    uint64_t nchannels, nbins;
    float predelay, max_time;     /* seconds: what fixPredelay subtracted (0 without trim_predelay); the latest arrival */
    uint64_t nimages;             /* merged image-source impulses that took part */
} rvb_pipeline_result;
int rvb_pipeline_create(rvb_pipeline ** out, rvb_ctx ** ctxs, uint64_t count, uint64_t group);
void rvb_pipeline_destroy(rvb_pipeline * p);
const char * rvb_pipeline_last_error(const rvb_pipeline * p);
int rvb_pipeline_configure_speakers(rvb_pipeline * p, const rvb_speaker * speakers, uint64_t nspeakers, int which, int remove_direct,
                                    int trim_predelay, float sample_rate, int mode, uint64_t nreflections, const float air_coefficient[8]);
int rvb_pipeline_configure_hrtf(rvb_pipeline * p, const float * table /* [2][360*180*8] */, const float facing[3], const float up[3],
                                int which, int remove_direct, int trim_predelay, float sample_rate, int mode, uint64_t nreflections,
                                const float air_coefficient[8]);
int rvb_pipeline_submit(rvb_pipeline * p, const float mic[3], const float source[3]);
int rvb_pipeline_submit_oriented(rvb_pipeline * p, const float mic[3], const float source[3], const float facing[3], const float up[3]);
uint64_t rvb_pipeline_pending(const rvb_pipeline * p);
int rvb_pipeline_next(rvb_pipeline * p, rvb_pipeline_result * out);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------
 * Durations in milliseconds of the kernels of the last rvb_trace / rvb_ir_accumulate, taken with
 * HIP events on the context's stream; names is a ';'-separated list matching ms[]. */
int rvb_last_timings(rvb_ctx * ctx, char * names, uint64_t names_capacity, float * ms, uint64_t ms_capacity, uint64_t * count);
/* In-kernel cycle stamps of the last trace; all zero unless the library was built with -DRVB_STAMPS=1
 * (diagnostic build, never the shipped one).  out[0..15] path_kernel, out[16..31] shadow_kernel. */
int rvb_debug_stamps(rvb_ctx * ctx, uint64_t * out, uint64_t capacity);
/* Number of bounces actually executed by the last trace (escaped rays stop early). */
int rvb_executed_bounces(rvb_ctx * ctx, uint64_t * bounces);

#ifdef __cplusplus
}
#endif
#endif
