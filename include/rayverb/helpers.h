// helpers.h — ray-direction generation, interface of reference rayverb/helpers.h:21-30.
#pragma once

#include "clstructs.h"

#include <vector>

// Point on the unit sphere, -1 <= z <= 1, -pi <= theta <= pi (reference helpers.cpp:63-67).
cl_float3 spherePoint(float z, float theta);

// Uniformly random unit vectors, seeded from the wall clock like the reference (helpers.cpp:69-81).
std::vector<cl_float3> getRandomDirections(unsigned long num);

// Same distribution from a fixed seed (reproducible runs; not in the reference).
std::vector<cl_float3> getSeededDirections(unsigned long num, unsigned long seed);
