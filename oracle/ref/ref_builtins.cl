// TEST INFRASTRUCTURE ONLY (oracle/_ref recipe) — never linked into the product.
//
// The reference hot path is OpenCL C (reference rayverb/kernel.cpp:9-627).  When
// that text is compiled for the x86 host it leaves 13 OpenCL 1.2 *language*
// built-ins undefined.  This file defines them, following the OpenCL 1.2
// specification (section 6.12) with the most literal evaluation order and no
// FMA contraction.  These definitions *define* the oracle:
//
//   dot(a,b)        = a.x*b.x + a.y*b.y + a.z*b.z          (left to right)
//   cross(a,b)      = (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx)
//   length(v)       = sqrt(dot(v,v))                         (correctly rounded sqrt)
//   distance(a,b)   = length(a - b)
//   normalize(v)    = v / length(v); v unchanged if length is 0 (OpenCL 1.2 §6.12.5;
//                     reference tests/attenuation_tests.h:29 relies on it)
//   all / any       = sign bit of every / any lane (relationals give -1 / 0)
//   pow, atan2      = correctly rounded (computed in double, rounded once)
//   degrees(x)      = x * (180/pi as float)
//   fabs            = sign-bit clear
//
// pow/atan2 precision is implementation-defined in OpenCL (16 / 6 ULP allowed);
// the correctly rounded value is inside every conforming implementation's range
// and is what lets the HIP path be compared bit for bit.

double rvb_ref_pow_d(double, double);      // libm, supplied by ref_harness.c
double rvb_ref_atan2_d(double, double);    // libm, supplied by ref_harness.c
float  rvb_ref_sqrtf(float);               // libm sqrtf (IEEE correctly rounded)

#ifdef RVB_REF_ALT_BUILTINS
// ---- a SECOND conforming choice (oracle/sensitivity.py): what a GPU OpenCL compiler typically emits -------------------
// dot / cross as fused multiply-adds (OpenCL C's FP_CONTRACT is ON by default), normalize as a multiply by the reciprocal
// square root, pow / atan2 in binary32 (libm powf / atan2f: within 1 ulp, not correctly rounded).  The kernel text itself is
// then compiled with -ffp-contract=fast as well (build_ref.sh alt).  Nothing here is used to define parity: it measures how
// many discrete outcomes move when the implementation-defined parts of OpenCL move.
float rvb_ref_powf(float, float);
float rvb_ref_atan2f(float, float);

float __attribute__((overloadable)) dot(float3 a, float3 b)
{
    return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z));
}

float3 __attribute__((overloadable)) cross(float3 a, float3 b)
{
    return (float3)(__builtin_fmaf(a.y, b.z, -(a.z * b.y)),
                    __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
                    __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}

float __attribute__((overloadable)) length(float3 v)
{
    return rvb_ref_sqrtf(__builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, v.z * v.z)));
}

float __attribute__((overloadable)) length(float2 v)
{
    return rvb_ref_sqrtf(__builtin_fmaf(v.x, v.x, v.y * v.y));
}

float __attribute__((overloadable)) distance(float3 a, float3 b)
{
    float3 d = a - b;
    return rvb_ref_sqrtf(__builtin_fmaf(d.x, d.x, __builtin_fmaf(d.y, d.y, d.z * d.z)));
}

float3 __attribute__((overloadable)) normalize(float3 v)
{
    float l2 = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, v.z * v.z));
    if (l2 == 0.0f)
        return v;
    float r = 1.0f / rvb_ref_sqrtf(l2);        // rsqrt, then three multiplies
    return (float3)(v.x * r, v.y * r, v.z * r);
}
#else
float __attribute__((overloadable)) dot(float3 a, float3 b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}

float3 __attribute__((overloadable)) cross(float3 a, float3 b)
{
    return (float3)(a.y * b.z - a.z * b.y,
                    a.z * b.x - a.x * b.z,
                    a.x * b.y - a.y * b.x);
}

float __attribute__((overloadable)) length(float3 v)
{
    return rvb_ref_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
}

float __attribute__((overloadable)) length(float2 v)
{
    return rvb_ref_sqrtf(v.x * v.x + v.y * v.y);
}

float __attribute__((overloadable)) distance(float3 a, float3 b)
{
    float3 d = a - b;
    return rvb_ref_sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
}

float3 __attribute__((overloadable)) normalize(float3 v)
{
    float l = rvb_ref_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (l == 0.0f)
        return v;
    return (float3)(v.x / l, v.y / l, v.z / l);
}

#endif

int __attribute__((overloadable)) all(int3 v)
{
    return (v.x < 0) && (v.y < 0) && (v.z < 0);
}

int __attribute__((overloadable)) any(int8 v)
{
    return (v.s0 < 0) || (v.s1 < 0) || (v.s2 < 0) || (v.s3 < 0)
        || (v.s4 < 0) || (v.s5 < 0) || (v.s6 < 0) || (v.s7 < 0);
}

#ifdef RVB_REF_ALT_BUILTINS
float8 __attribute__((overloadable)) pow(float8 x, float8 y)
{
    float8 r;
    r.s0 = rvb_ref_powf(x.s0, y.s0); r.s1 = rvb_ref_powf(x.s1, y.s1); r.s2 = rvb_ref_powf(x.s2, y.s2); r.s3 = rvb_ref_powf(x.s3, y.s3);
    r.s4 = rvb_ref_powf(x.s4, y.s4); r.s5 = rvb_ref_powf(x.s5, y.s5); r.s6 = rvb_ref_powf(x.s6, y.s6); r.s7 = rvb_ref_powf(x.s7, y.s7);
    return r;
}
#else
float8 __attribute__((overloadable)) pow(float8 x, float8 y)
{
    float8 r;
    r.s0 = (float) rvb_ref_pow_d((double) x.s0, (double) y.s0);
    r.s1 = (float) rvb_ref_pow_d((double) x.s1, (double) y.s1);
    r.s2 = (float) rvb_ref_pow_d((double) x.s2, (double) y.s2);
    r.s3 = (float) rvb_ref_pow_d((double) x.s3, (double) y.s3);
    r.s4 = (float) rvb_ref_pow_d((double) x.s4, (double) y.s4);
    r.s5 = (float) rvb_ref_pow_d((double) x.s5, (double) y.s5);
    r.s6 = (float) rvb_ref_pow_d((double) x.s6, (double) y.s6);
    r.s7 = (float) rvb_ref_pow_d((double) x.s7, (double) y.s7);
    return r;
}

#endif

float __attribute__((overloadable)) fabs(float x)
{
    return as_float(as_uint(x) & 0x7fffffffu);
}

#ifdef RVB_REF_ALT_BUILTINS
float __attribute__((overloadable)) atan2(float y, float x)
{
    return rvb_ref_atan2f(y, x);
}
#else
float __attribute__((overloadable)) atan2(float y, float x)
{
    return (float) rvb_ref_atan2_d((double) y, (double) x);
}

#endif

float __attribute__((overloadable)) degrees(float x)
{
    return x * 57.295779513082320877f;
}
