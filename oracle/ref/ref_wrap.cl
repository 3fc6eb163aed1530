// TEST INFRASTRUCTURE ONLY (oracle/_ref recipe) — never linked into the product.
//
// Entry points that let a C harness call the reference's three OpenCL kernels
// (reference rayverb/kernel.cpp:304, :515, :586) after that text has been
// compiled for the host.  "rvb_ref_kernel_text.cl" is produced at build time in
// a temporary directory by oracle/ref/build_ref.sh from the text where it lies
// under /root/reference; it is never written into the repository.
//
// Arguments travel through plain structs of pointers / scalars / float arrays so
// that no OpenCL vector type crosses the C <-> OpenCL-C boundary.

#include "rvb_ref_kernel_text.cl"

typedef struct {
    global float3 * directions;
    global Triangle * triangles;
    unsigned long numtriangles;
    global float3 * vertices;
    global Surface * surfaces;
    global Impulse * impulses;
    global Impulse * image_source;
    global unsigned long * image_source_index;
    unsigned long outputOffset;
    float position [3];
    float source [3];
    float air [8];
} RvbRefRaytraceArgs;

void rvb_ref_run_raytrace (global RvbRefRaytraceArgs * a);
void rvb_ref_run_raytrace (global RvbRefRaytraceArgs * a)
{
    raytrace
    (   a->directions
    ,   (float3) (a->position [0], a->position [1], a->position [2])
    ,   a->triangles
    ,   a->numtriangles
    ,   a->vertices
    ,   (float3) (a->source [0], a->source [1], a->source [2])
    ,   a->surfaces
    ,   a->impulses
    ,   a->image_source
    ,   a->image_source_index
    ,   a->outputOffset
    ,   (float8) (a->air [0], a->air [1], a->air [2], a->air [3],
                  a->air [4], a->air [5], a->air [6], a->air [7])
    );
}

typedef struct {
    global Impulse * in;
    global AttenuatedImpulse * out;
    float mic [3];
    float direction [3];
    float coefficient;
} RvbRefAttenuateArgs;

void rvb_ref_run_attenuate (global RvbRefAttenuateArgs * a);
void rvb_ref_run_attenuate (global RvbRefAttenuateArgs * a)
{
    Speaker s;
    s.direction = (float3) (a->direction [0], a->direction [1], a->direction [2]);
    s.coefficient = a->coefficient;
    attenuate
    (   (float3) (a->mic [0], a->mic [1], a->mic [2])
    ,   a->in
    ,   a->out
    ,   s
    );
}

typedef struct {
    global Impulse * in;
    global AttenuatedImpulse * out;
    global VolumeType * table;
    float mic [3];
    float pointing [3];
    float up [3];
    unsigned long channel;
} RvbRefHrtfArgs;

void rvb_ref_run_hrtf (global RvbRefHrtfArgs * a);
void rvb_ref_run_hrtf (global RvbRefHrtfArgs * a)
{
    hrtf
    (   (float3) (a->mic [0], a->mic [1], a->mic [2])
    ,   a->in
    ,   a->out
    ,   a->table
    ,   (float3) (a->pointing [0], a->pointing [1], a->pointing [2])
    ,   (float3) (a->up [0], a->up [1], a->up [2])
    ,   a->channel
    );
}
