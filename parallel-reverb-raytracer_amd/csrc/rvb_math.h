// rvb_math.h — the geometry arithmetic of the hot path, one IEEE binary32 operation per
// operator in the order written (compile with -ffp-contract=off; sqrt and divide correctly
// rounded, which hipcc does by default for HIP).  Host and device share these definitions so
// that data precomputed on the host at rvb_set_scene (edges, normals) is bit-identical to what
// the kernels would compute on the fly.
//
// Parity notes (reference rayverb/kernel.cpp): the OpenCL built-ins dot / cross / length /
// normalize are implementation-defined in precision; here they are the literal left-to-right
// forms, normalize(0) = 0 (OpenCL 1.2 §6.12.5, relied on by reference
// tests/attenuation_tests.h:29), and pow / atan2 are evaluated in binary64 and rounded once
// (correctly rounded with overwhelming probability) — see DESIGN.md "Arithmetic contract".
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define RVB_HD __host__ __device__ __forceinline__

#define RVB_EPSILON 0.0001f                         // reference kernel.cpp:11
#define RVB_NUM_IMAGE_SOURCE 10                     // reference clstructs.h:4

struct v3 { float x, y, z; };

RVB_HD v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RVB_HD v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RVB_HD v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RVB_HD v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RVB_HD v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
RVB_HD float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RVB_HD v3 cross3(v3 a, v3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RVB_HD float length3(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RVB_HD v3 normalize3(v3 a)
{
    float l = length3(a);
    if (l == 0.0f)
        return a;
    return mk3(a.x / l, a.y / l, a.z / l);
}

// reference kernel.cpp:14: constant float SECONDS_PER_METER = 1.0f / 340.000000 (a double quotient)
RVB_HD float seconds_per_meter() { return (float) (1.0 / 340.000000); }

// A triangle as the kernels see it: first vertex and the two edges of reference
// kernel.cpp:65-66 (e0 = v1 - v0, e1 = v2 - v0), precomputed with the same subtraction.
struct TriEdges { v3 v0, e0, e1; };

// 1.0f / x, correctly rounded.  On the device: v_rcp_f32 (1 ulp) + one FMA Newton step — three instructions instead of the
// ten of the general division sequence.  tools/rcp_probe.hip compares it with `1.0f / x` for EVERY float with
// 2^-17 <= |x| <= 2^64, both signs (1 358 954 498 values): no difference on gfx950.  mt_intersect only divides by
// determinants with |det| >= EPSILON = 1e-4 > 2^-17, and rvb_build_scene rejects coordinates beyond 2^30, so |det| < 2^64.
RVB_HD float reciprocal_cr(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e0 = fmaf(-x, r0, 1.0f);
    return fmaf(e0, r0, r0);
#else
    return 1.0f / x;
#endif
}

// reference kernel.cpp:62-88 (triangle_vert_intersection).  Returns 0 for "no hit".
RVB_HD float mt_intersect(const v3 & v0, const v3 & e0, const v3 & e1, const v3 & pos, const v3 & dir)
{
    v3 pvec = cross3(dir, e1);
    float det = dot3(e0, pvec);
    if (-RVB_EPSILON < det && det < RVB_EPSILON)
        return 0.0f;
    float invdet = reciprocal_cr(det);
    v3 tvec = pos - v0;
    float ucomp = invdet * dot3(tvec, pvec);
    if (ucomp < 0.0f || 1.0f < ucomp)
        return 0.0f;
    v3 qvec = cross3(tvec, e0);
    float vcomp = invdet * dot3(dir, qvec);
    if (vcomp < 0.0f || 1.0f < vcomp + ucomp)
        return 0.0f;
    return invdet * dot3(e1, qvec);
}

// Three stored vertices (image-source mirrored triangles, reference kernel.cpp:56-60).
struct TriVerts { v3 v0, v1, v2; };

RVB_HD float mt_intersect_verts(const TriVerts & t, const v3 & pos, const v3 & dir)
{
    return mt_intersect(t.v0, t.v1 - t.v0, t.v2 - t.v0, pos, dir);
}

// reference kernel.cpp:109-116
RVB_HD v3 verts_normal(const TriVerts & t) { return normalize3(cross3(t.v1 - t.v0, t.v2 - t.v0)); }

// reference kernel.cpp:129-133: direction - (normal * 2 * dot(direction, normal))
RVB_HD v3 reflect3(v3 normal, v3 direction)
{
    v3 n2 = normal * 2.0f;
    float d = dot3(direction, normal);
    return direction - n2 * d;
}

// reference kernel.cpp:216-221: *p += -n * dot(n, *p - t->v0) * 2
RVB_HD void mirror_point(v3 & p, const TriVerts & t)
{
    v3 n = verts_normal(t);
    float d = dot3(n, p - t.v0);
    p = p + ((-n) * d) * 2.0f;
}

// reference kernel.cpp:223-229
RVB_HD void mirror_verts(TriVerts & in, const TriVerts & t)
{
    mirror_point(in.v0, t);
    mirror_point(in.v1, t);
    mirror_point(in.v2, t);
}

// reference kernel.cpp:194-198: pow(M_E, distance * AIR) with M_E converted to float.  The
// correctly rounded float result is wanted (the oracle's definition of the OpenCL built-in).
// pow(e_f, x) = exp(x * ln(e_f)): x is a float, so x * ln(e_f) is formed exactly-to-106-bits with
// ln(e_f) split in two doubles, and exp is evaluated in binary64; the total error is ~1.5 ulp of
// binary64, so rounding to float gives the correctly rounded value except with probability ~1e-8.
// The "* 1" of kernel.cpp:210-213 is exact.
RVB_HD float air_attenuation(float distance, float air)
{
#if defined(RVB_EXP_PROBE)          // diagnostic builds only (wrong values): what the kernels cost WITHOUT the exponential
    return 1.0f + distance * air;
#endif
    // ln((float) M_E) = ln(2.71828174591064453125) = 0.99999996963214001827215631464059433...
    const double ln_e_hi = 0x1.fffffefb245eap-1;
    const double ln_e_lo = 0x1.a2d208d1c4e82p-56;      // ln(e_f) - ln_e_hi
    const double x = (double) (distance * air);
    const double p = x * ln_e_hi;
    const double r = fma(x, ln_e_hi, -p) + x * ln_e_lo; // exact rounding error of p plus the low part
    const double e = exp(p);
    return (float) (e + e * r);
}
