// cl.hpp — callers of the reference include the Khronos OpenCL C++ bindings for cl_float3 /
// cl_float8 / cl_ulong / cl::Error only; those come from rayverb/cl_compat.h here (no OpenCL).
#pragma once
#include "rayverb/cl_compat.h"
