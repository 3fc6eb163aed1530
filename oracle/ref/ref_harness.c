/* TEST INFRASTRUCTURE ONLY (oracle/_ref recipe) — never linked into the product.
 *
 * C harness around the host-compiled reference kernels.  It plays the part of
 * the OpenCL runtime for them: it supplies get_global_id, the three libm calls
 * the built-in definitions in ref_builtins.cl use, and loops over the NDRange
 * the way reference rayverb/rayverb.cpp:619-642, :878-884, :801-810 enqueue it.
 *
 * Exported (ctypes) entry points: rvb_ref_raytrace, rvb_ref_attenuate,
 * rvb_ref_hrtf.  Buffers are zero-filled before the raytrace kernel runs, as
 * reference rayverb/rayverb.cpp:600-616 does for every group.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t surface, v0, v1, v2; } RefTriangle;                 /* 32 B */
typedef struct { float specular[8]; float diffuse[8]; } __attribute__((aligned(32))) RefSurface; /* 64 B */
typedef struct { float volume[8]; float position[4]; float time; float pad[3]; } __attribute__((aligned(32))) RefImpulse; /* 64 B */
typedef struct { float volume[8]; float time; float pad[7]; } __attribute__((aligned(32))) RefAttenuated; /* 64 B */

_Static_assert(sizeof(RefTriangle) == 32, "Triangle layout");
_Static_assert(sizeof(RefSurface) == 64, "Surface layout");
_Static_assert(sizeof(RefImpulse) == 64, "Impulse layout");
_Static_assert(sizeof(RefAttenuated) == 64, "AttenuatedImpulse layout");

typedef struct {
    void * directions;
    RefTriangle * triangles;
    uint64_t numtriangles;
    void * vertices;
    RefSurface * surfaces;
    RefImpulse * impulses;
    RefImpulse * image_source;
    uint64_t * image_source_index;
    uint64_t outputOffset;
    float position[3];
    float source[3];
    float air[8];
} RefRaytraceArgs;

typedef struct {
    RefImpulse * in;
    RefAttenuated * out;
    float mic[3];
    float direction[3];
    float coefficient;
} RefAttenuateArgs;

typedef struct {
    RefImpulse * in;
    RefAttenuated * out;
    void * table;
    float mic[3];
    float pointing[3];
    float up[3];
    uint64_t channel;
} RefHrtfArgs;

void rvb_ref_run_raytrace(RefRaytraceArgs *);
void rvb_ref_run_attenuate(RefAttenuateArgs *);
void rvb_ref_run_hrtf(RefHrtfArgs *);

/* --- what the OpenCL runtime would provide ------------------------------- */
static _Thread_local size_t g_global_id;
size_t _Z13get_global_idj(unsigned dim) { (void) dim; return g_global_id; }

double rvb_ref_pow_d(double x, double y) { return pow(x, y); }
double rvb_ref_atan2_d(double y, double x) { return atan2(y, x); }
float rvb_ref_sqrtf(float x) { return sqrtf(x); }
float rvb_ref_powf(float x, float y) { return powf(x, y); }          /* the alternative built-ins only (oracle/sensitivity.py) */
float rvb_ref_atan2f(float y, float x) { return atan2f(y, x); }

/* --- NDRange loops ------------------------------------------------------- */
void rvb_ref_raytrace
(   const float * directions      /* [nrays][4] */
,   uint64_t nrays
,   const RefTriangle * triangles
,   uint64_t ntriangles
,   const float * vertices        /* [nverts][4] */
,   const RefSurface * surfaces
,   const float mic[3]
,   const float source[3]
,   uint64_t nreflections
,   const float air[8]
,   RefImpulse * impulses         /* [nrays * nreflections] */
,   RefImpulse * image_source     /* [nrays * 10] */
,   uint64_t * image_source_index /* [nrays * 10] */
)
{
    memset(impulses, 0, sizeof(RefImpulse) * nrays * nreflections);
    memset(image_source, 0, sizeof(RefImpulse) * nrays * 10);
    memset(image_source_index, 0, sizeof(uint64_t) * nrays * 10);

    RefRaytraceArgs a;
    a.directions = (void *) directions;
    a.triangles = (RefTriangle *) triangles;
    a.numtriangles = ntriangles;
    a.vertices = (void *) vertices;
    a.surfaces = (RefSurface *) surfaces;
    a.impulses = impulses;
    a.image_source = image_source;
    a.image_source_index = image_source_index;
    a.outputOffset = nreflections;
    for (int i = 0; i != 3; ++i) { a.position[i] = mic[i]; a.source[i] = source[i]; }
    for (int i = 0; i != 8; ++i) a.air[i] = air[i];

    #pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t) nrays; ++i)
    {
        RefRaytraceArgs local = a;
        g_global_id = (size_t) i;
        rvb_ref_run_raytrace(&local);
    }
}

void rvb_ref_attenuate
(   const float mic[3]
,   const RefImpulse * in
,   uint64_t n
,   const float direction[3]
,   float coefficient
,   RefAttenuated * out           /* caller pre-fills (reference leaves untouched slots alone) */
)
{
    RefAttenuateArgs a;
    a.in = (RefImpulse *) in;
    a.out = out;
    for (int i = 0; i != 3; ++i) { a.mic[i] = mic[i]; a.direction[i] = direction[i]; }
    a.coefficient = coefficient;
    for (uint64_t i = 0; i != n; ++i)
    {
        g_global_id = (size_t) i;
        rvb_ref_run_attenuate(&a);
    }
}

void rvb_ref_hrtf
(   const float mic[3]
,   const RefImpulse * in
,   uint64_t n
,   const float * table           /* [360*180][8], 32-byte aligned */
,   const float pointing[3]
,   const float up[3]
,   uint64_t channel
,   RefAttenuated * out
)
{
    RefHrtfArgs a;
    a.in = (RefImpulse *) in;
    a.out = out;
    a.table = (void *) table;
    for (int i = 0; i != 3; ++i) { a.mic[i] = mic[i]; a.pointing[i] = pointing[i]; a.up[i] = up[i]; }
    a.channel = channel;
    for (uint64_t i = 0; i != n; ++i)
    {
        g_global_id = (size_t) i;
        rvb_ref_run_hrtf(&a);
    }
}
