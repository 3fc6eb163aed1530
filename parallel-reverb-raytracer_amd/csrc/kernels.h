// kernels.h — launch interface between the C-ABI (capi.hip) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rvb_capi.h"
#include "bvh.h"

struct SceneDev {                       // device pointers of the current scene
    const BvhNode * nodes = nullptr;
    const BvhTri * tris = nullptr;
    const TriShade * shade = nullptr;
    const TriCorners * corners = nullptr;
    const rvb_surface * surfaces = nullptr;
    uint32_t ntris = 0;                  // entries of tris[]
    float cull_abs = 0.0f;              // slack added to the running closest distance when culling
    float cull_rel = 0.0f;
    unsigned long long * stamps = nullptr;   // diagnostic builds only (RVB_STAMPS)
};

// entry of the image-source work list: ray within the launch and bounce; index 0xFFFFFFFF = the direct path of pair `ray`
struct ImageItem { uint32_t ray, index; };
struct TraceArgs {
    SceneDev scene;
    const float4 * directions;          // [nrays]
    rvb_impulse * impulses;             // [nrays * nreflections]
    uint32_t * early;                   // [nrays * 9] triangle hit at bounce 0..8, 0xFFFFFFFF = none
    rvb_image_candidate * candidates;   // [nrays * 9] capacity
    uint32_t * candidate_count;
    ImageItem * image_items;     // [nrays * 9 + npairs] (ray, bounce) pairs whose image ray crosses every image triangle (image_plan_kernel)
    uint32_t * image_item_count;
    uint32_t * image_state;             // [capacity of image_items] per listed pair: queries done (low half), queries failed (high half)
    rvb_impulse * direct;               // slot 0
    unsigned long long * executed;      // bounces executed
    uint32_t * sort_keys;               // [nrays * nreflections] leaf position of the triangle hit, 0xFFFFFFFF = no record (or null)
    uint16_t * sort_keys16;             // the same as 16-bit keys (leaf position >> key_shift, 0xFFFF = no record), written in 64-byte runs through LDS
                                        // (path kernels; needs nreflections % 32 == 0); exactly one of sort_keys / sort_keys16 is set, or neither
    uint32_t key_shift;
    uint32_t * sort_order;              // [nrays * nreflections] record indices grouped by bucket
    uint32_t * time_range;              // [2] float bits: min non-zero / max time of non-zero diffuse impulses
    uint64_t nrays;
    uint32_t nreflections;
    uint32_t stack_entries;             // LDS traversal stack entries per lane (BuiltScene::stack_need)
    uint32_t lds_surfaces;              // surfaces staged in LDS behind the stack by the quad kernels (rvb_lds_surfaces), 0 = none
    uint32_t scene_nodes;               // number of BVH nodes (experiments that stage the top of the tree)
    uint32_t path_lanes;                // lanes per ray in the path kernel: 4 (path_kernel) or 2 (path_pair_kernel), rvb_path_lanes_for
    // Several (source, microphone) pairs in ONE launch (rvb_trace_pairs): ray r belongs to pair r / rays_per_pair and uses
    // direction r % rays_per_pair; its records, early ids, candidates follow the global ray number.  npairs == 1: mic / source below.
    uint32_t npairs;
    uint32_t rays_per_pair;
    const float4 * pair_mics;           // device [npairs] (xyz, w unused); direct[] and time_range[] then hold one entry per pair
    const float4 * pair_sources;
    uint64_t ray_offset;
    float mic[3];
    float source[3];
    float air[8];
};

// Phase A: one lane per ray, the sequential closest-hit / reflect chain (kernel.cpp:359-375,
// :459-461, :478, :492-501).  Leaves a work record per bounce in impulses[].
void rvb_launch_path(const TraceArgs & a, hipStream_t s);
// How many surfaces the quad kernels stage in LDS for this scene (all of them, or 0 when they would cost occupancy).
uint32_t rvb_lds_surfaces(uint32_t stack_entries, uint64_t nsurfaces);
// lanes per ray for a launch of `nrays` rays when the caller keeps `concurrent` such traces in flight on the device
uint32_t rvb_path_lanes_for(uint64_t nrays, uint32_t concurrent);
// the two-lane path kernel for up to RVB_MAX_GROUP traces (contexts) in ONE launch: their waves are dispatched and scheduled together
#define RVB_MAX_GROUP 4
void rvb_launch_path_group(const TraceArgs * traces, uint32_t count, hipStream_t s);
uint32_t rvb_shadow_lanes();        // lanes per record in the shadow kernel: 2 (shadow_pair_kernel) unless RVB_SHADOW_LANES=4
// Phase C: image-source validation (kernel.cpp:379-457) + slot 0: a plan kernel (one lane per ray) and a check kernel (four lanes per listed pair).
void rvb_launch_images(const TraceArgs & a, hipStream_t s);
// Phase B: one lane per (ray, bounce): diffuse shadow ray to the microphone and the final
// Impulse (kernel.cpp:463-490).  Overwrites the work records.
void rvb_launch_shadow(const TraceArgs & a, hipStream_t s);
// Grouping of the work records by the leaf position of the triangle they start from (stream_kernels.hip):
// order[] lists the records bucket by bucket and the shadow kernel walks that list.
size_t rvb_group_records_temp_bytes(uint64_t n);
hipError_t rvb_group_records(void * temp, size_t temp_bytes, const uint32_t * keys, uint32_t * keys_scratch, uint32_t * order,
                             uint64_t n, uint32_t first_record, int begin_bit, int end_bit, hipStream_t s);
hipError_t rvb_group_records16(void * temp, size_t temp_bytes, const uint16_t * keys, uint16_t * keys_scratch, uint32_t * order,
                               uint64_t n, uint32_t first_record, int begin_bit, int end_bit, hipStream_t s);

// ---- streaming kernels (stream_kernels.hip) ---------------------------------------------------
struct AttenuationModel {
    int hrtf = 0;                       // 0: speakers, 1: hrtf
    uint32_t nchannels = 0;             // speakers: <= 8; hrtf: 2
    float mic[3] = {0, 0, 0};
    float speaker_dir[8][3] = {};       // normalised on device exactly as kernel.cpp:511 does
    float speaker_coeff[8] = {};
    const float * hrtf_table = nullptr; // device [2][360*180+1][8]
    float facing[3] = {0, 0, 0}, up[3] = {0, 0, 0};
};

// kernel attenuate / hrtf for one channel, materialised (kernel.cpp:515-535, :586-625)
void rvb_launch_attenuate(const AttenuationModel & m, uint32_t channel, const rvb_impulse * in, uint64_t n,
                          rvb_attenuated_impulse * out, hipStream_t s);
// min non-zero / max attenuated time over all channels -> range[0], range[1] (uint bits of floats;
// caller initialises to 0xFFFFFFFF / 0)
void rvb_launch_time_range(const AttenuationModel & m, const rvb_impulse * in, uint64_t n, uint32_t * range, hipStream_t s);
// fused attenuate + predelay + bin with float atomics into the image acc[nbins][nchannels][8]
void rvb_launch_histogram_fast(const AttenuationModel & m, const rvb_impulse * in, uint64_t n, float predelay,
                               float sample_rate, uint64_t nbins, float * acc, hipStream_t s);
// acc[bin][channel][band] -> hist[channel][band][bin] (added to what is there)
void rvb_launch_histogram_transpose(const float * acc, float * hist, uint32_t nchannels, uint64_t nbins, hipStream_t s);
// exact mode helpers: per-impulse bin keys (`sentinel` = nbins marks impulses that add nothing), then the ordered per-bin
// summation of `nchannels` channels that share the sorted list, added to what hist [all channels][8][nbins] holds
void rvb_launch_bin_keys(const AttenuationModel & m, uint32_t channel, const rvb_impulse * in, uint64_t n, uint64_t index_base,
                         float predelay, float sample_rate, uint32_t sentinel, uint32_t * keys, uint32_t * values, hipStream_t s);
void rvb_launch_ordered_sum(const AttenuationModel & m, uint32_t first_channel, uint32_t nchannels, const rvb_impulse * diffuse,
                            uint64_t ndiffuse, const rvb_impulse * images, uint64_t nimages,
                            const uint32_t * sorted_values, const uint32_t * starts, const uint32_t * ends, uint64_t n,
                            uint64_t nbins, float * hist, hipStream_t s, uint64_t bin_begin = 0, uint64_t bin_end = ~0ull);   // bins [bin_begin, bin_end) only
// HRTF model, both ears at once: ONE list of 2 n (key, value) entries — ear e's entry of impulse j at e * n + j, key e * (nbins + 1) + bin
// (sentinel e * (nbins + 1) + nbins) — and the ordered sum of both ears over its sorted form (starts / ends indexed by that key)
void rvb_launch_bin_keys_hrtf(const AttenuationModel & m, const rvb_impulse * in, uint64_t count, uint64_t index_base, uint64_t n,
                              float predelay, float sample_rate, uint32_t nbins, uint32_t * keys, uint32_t * values, hipStream_t s);
void rvb_launch_ordered_sum_hrtf(const AttenuationModel & m, const rvb_impulse * diffuse, uint64_t ndiffuse, const rvb_impulse * images,
                                 const uint32_t * sorted_values, const uint32_t * starts, const uint32_t * ends, uint64_t nbins, float * hist,
                                 hipStream_t s, uint64_t bin_begin = 0, uint64_t bin_end = ~0ull);
// starts[key] / ends[key] = first position of `key` in the sorted list / one past its last (starts[] pre-filled with 0xFFFFFFFF by
// the caller, nbins entries each; ends[] is only read where starts[] was written)
void rvb_launch_bin_starts(const uint32_t * sorted_keys, uint64_t n, uint64_t nbins, uint32_t * starts, uint32_t * ends, hipStream_t s);
// flattenImpulses of already attenuated impulses (rayverb.cpp:48-77): keys + ordered sum
void rvb_launch_flat_keys(const rvb_attenuated_impulse * in, uint64_t n, float sample_rate, uint32_t * keys, uint32_t * values,
                          uint32_t * max_time_bits, hipStream_t s);
void rvb_launch_flat_ordered_sum(const rvb_attenuated_impulse * in, const uint32_t * sorted_keys, const uint32_t * sorted_values,
                                 const uint32_t * starts, uint64_t n, uint64_t nbins, float * out, hipStream_t s);
void rvb_launch_fix_predelay(rvb_attenuated_impulse * a, uint64_t n, float seconds, hipStream_t s);
// stable sort of (key, value) pairs by key (device radix sort); temp storage managed by the caller
// csrc/radix_sort.hip: stable LSD radix sort whose kernels fit beside resident path waves (single-wave workgroups, <= 32 VGPRs)
size_t rvb_radix_sort_temp_bytes(uint64_t n);
hipError_t rvb_radix_sort_pairs(void * temp, size_t temp_bytes, const uint32_t * keys, const uint32_t * values, uint32_t value_base,
                                uint32_t * keys_a, uint32_t * values_a, uint32_t * keys_b, uint32_t * values_b, uint64_t n,
                                int begin_bit, int end_bit, bool want_keys, const uint32_t ** keys_sorted, const uint32_t ** values_sorted,
                                hipStream_t s);
size_t rvb_sort_temp_bytes(uint64_t n);
void rvb_sort_pairs(void * temp, size_t temp_bytes, const uint32_t * keys_in, uint32_t * keys_out,
                    const uint32_t * values_in, uint32_t * values_out, uint64_t n, int key_bits, hipStream_t s);
