/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's per-ray hot path.  Nothing in the
 * product may include, link or call this: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do (as the checker / the timed CPU baseline).
 *
 * Parity pinning: this restatement is checked (tests/test_oracle_*.py) against
 *   - the known answers of the reference's own gtest suites
 *     (reference tests/raytrace_tests.h:35-47, tests/attenuation_tests.h:67-101,
 *      tests/hrtf_tests.cpp:42-85), and
 *   - golden vectors generated in the build container from the reference's own
 *     kernel text compiled for the host (oracle/ref/build_ref.sh, tests/golden/).
 *
 * All arithmetic is IEEE binary32, evaluated in source order with no FMA
 * contraction; sqrt and divide are correctly rounded; pow and atan2 are correctly
 * rounded (evaluated in binary64 and rounded once) — the same definitions that
 * oracle/ref/ref_builtins.cl gives the reference kernel's OpenCL built-ins.
 */
#ifndef RVB_ORACLE_H
#define RVB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVBO_NUM_IMAGE_SOURCE 10        /* reference rayverb/clstructs.h:4 */
#define RVBO_SPEED_OF_SOUND 340.0f      /* reference rayverb/clstructs.h:5 */
#define RVBO_EPSILON 0.0001f            /* reference rayverb/kernel.cpp:11 */

/* Layout-identical to reference rayverb/clstructs.h:17-58 (sizes: SURVEY.md §8(a)). */
typedef struct { uint64_t surface, v0, v1, v2; } RvboTriangle;                                    /* 32 B */
typedef struct { float specular[8]; float diffuse[8]; } __attribute__((aligned(32))) RvboSurface;   /* 64 B */
typedef struct { float volume[8]; float position[4]; float time; float pad[3]; }
    __attribute__((aligned(32))) RvboImpulse;                                                     /* 64 B */
typedef struct { float volume[8]; float time; float pad[7]; }
    __attribute__((aligned(32))) RvboAttenuated;                                                  /* 64 B */

/* reference rayverb/kernel.cpp:304-503 (kernel raytrace) driven the way
 * rayverb/rayverb.cpp:587-684 drives it, for exactly nrays rays (quirk Q1: the
 * reference's stale work-items of a partial last group are not reproduced).
 * Output buffers are zero-filled first (rayverb/rayverb.cpp:600-616).
 * nthreads <= 0: use all cores. */
void rvbo_raytrace
(   const float * directions, uint64_t nrays,
    const RvboTriangle * triangles, uint64_t ntriangles,
    const float * vertices,
    const RvboSurface * surfaces,
    const float mic[3], const float source[3],
    uint64_t nreflections, const float air[8],
    RvboImpulse * impulses, RvboImpulse * image_source, uint64_t * image_source_index,
    int nthreads);

/* reference rayverb/kernel.cpp:167-192 / :274-296, exposed for BVH-vs-brute-force tests.
 * Returns 1 and fills primitive/distance when something is hit. */
int rvbo_closest_hit
(   const float origin[3], const float direction[3],
    const RvboTriangle * triangles, uint64_t ntriangles, const float * vertices,
    uint64_t * primitive, float * distance);
int rvbo_point_visible
(   const float begin[3], const float point[3],
    const RvboTriangle * triangles, uint64_t ntriangles, const float * vertices);

/* Host de-dup of image-source contributions, reference rayverb/rayverb.cpp:654-676,
 * followed by getRawImages, rayverb/rayverb.cpp:692-706.  Writes at most max_out
 * impulses in std::map key order; returns the count. */
uint64_t rvbo_collect_images
(   const RvboImpulse * image_source, const uint64_t * image_source_index, uint64_t nrays,
    int remove_direct, RvboImpulse * out, uint64_t max_out);

/* reference rayverb/kernel.cpp:505-535 (kernel attenuate).  Zero-volume impulses
 * produce {0, 0} (quirk Q2: the reference leaves them uninitialised). */
void rvbo_attenuate_speaker
(   const float mic[3], const RvboImpulse * in, uint64_t n,
    const float direction[3], float coefficient, RvboAttenuated * out);

/* reference rayverb/kernel.cpp:537-625 (kernel hrtf); table is [360*180 (+1)][8]. */
void rvbo_attenuate_hrtf
(   const float mic[3], const RvboImpulse * in, uint64_t n,
    const float * table, const float pointing[3], const float up[3],
    uint64_t channel, RvboAttenuated * out);
/* Selected table index a*180+e for one direction (pins quirk Q5). */
int64_t rvbo_hrtf_index(const float pointing[3], const float up[3], const float direction[3]);

/* reference rayverb/rayverb.h:49-97 over all channels. */
float rvbo_find_predelay(const RvboAttenuated * const * channels, uint64_t nchannels, uint64_t n);
void rvbo_fix_predelay(RvboAttenuated * impulses, uint64_t n, float seconds);

/* reference rayverb/rayverb.cpp:48-77.  rvbo_flatten_bins gives MAX_SAMPLE;
 * rvbo_flatten fills out[8][nbins] (zeroed first), serial impulse order. */
uint64_t rvbo_flatten_bins(const RvboAttenuated * impulses, uint64_t n, float samplerate);
void rvbo_flatten(const RvboAttenuated * impulses, uint64_t n, float samplerate, float * out, uint64_t nbins);

#ifdef __cplusplus
}
#endif
#endif
