// cl_compat.h — the handful of OpenCL host types that leak into the reference's public API
// (reference include/cl.hpp via rayverb/clstructs.h:1-2), re-declared layout-compatibly WITHOUT
// the OpenCL runtime: cl_float3 is a 16-byte float4, cl_float8 a 32-byte float8, both with the
// `.s[]` member the reference's callers index (e.g. cmd/main.cpp:142, tests/raytrace_tests.cpp:23).
// cl::Error exists because callers catch it (cmd/main.cpp:299-305); here it carries rvb_* codes.
#pragma once

#include <cstdint>
#include <exception>
#include <string>

typedef float cl_float;
typedef std::uint64_t cl_ulong;
typedef std::int32_t cl_int;

typedef union {
    cl_float s[4];
    struct { cl_float x, y, z, w; };
} __attribute__((aligned(16))) cl_float4;
typedef cl_float4 cl_float3;

typedef union {
    cl_float s[8];
    struct { cl_float s0, s1, s2, s3, s4, s5, s6, s7; };
} __attribute__((aligned(32))) cl_float8;

static_assert(sizeof(cl_float3) == 16 && alignof(cl_float3) == 16, "cl_float3 layout");
static_assert(sizeof(cl_float8) == 32 && alignof(cl_float8) == 32, "cl_float8 layout");

namespace cl {
class Error : public std::exception {
public:
    Error(cl_int err, const char * what = nullptr) : err_(err), what_(what ? what : "rvb error") {}
    virtual ~Error() throw() {}
    virtual const char * what() const throw() { return what_.c_str(); }
    cl_int err() const { return err_; }
private:
    cl_int err_;
    std::string what_;
};
}  // namespace cl
