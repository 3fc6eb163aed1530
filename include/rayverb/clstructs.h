// clstructs.h — data contracts of the ray tracer, names and layouts as reference
// rayverb/clstructs.h:4-58 (sizes: SURVEY.md §8(a) T1-T7).  The HIP kernels read and write these
// exact layouts; the static_asserts below are the contract.
#pragma once

#include "cl_compat.h"

#define NUM_IMAGE_SOURCE 10
#define SPEED_OF_SOUND (340.0f)

// 8 octave bands, low to high
typedef cl_float8 VolumeType;

struct Triangle {                // indices into the surface and vertex arrays
    cl_ulong surface;
    cl_ulong v0;
    cl_ulong v1;
    cl_ulong v2;
};

struct Surface {                 // per-band reflection coefficients
    VolumeType specular;
    VolumeType diffuse;
};

struct Impulse {                 // one contribution: 8-band volume, where it came from, arrival time
    VolumeType volume;
    cl_float3 position;
    cl_float time;
};

struct AttenuatedImpulse {       // after microphone / HRTF attenuation
    VolumeType volume;
    cl_float time;
};

struct Speaker {                 // polar pattern: 0 = omni ... 1 = bidirectional
    cl_float3 direction;
    cl_float coefficient;
};

static_assert(sizeof(Triangle) == 32, "Triangle layout");
static_assert(sizeof(Surface) == 64, "Surface layout");
static_assert(sizeof(Impulse) == 64, "Impulse layout");
static_assert(sizeof(AttenuatedImpulse) == 64, "AttenuatedImpulse layout");
static_assert(sizeof(Speaker) == 32, "Speaker layout");
