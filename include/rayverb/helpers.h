// helpers.h — ray-direction generation and the diagnostic dump, interface of reference rayverb/helpers.h:7-30.
#pragma once

#include "clstructs.h"

#include <string>
#include <vector>

// One line of JSON per ray — [{"position":[x,y,z],"volume":mean of the 8 bands}, ...], one object per
// reflection — the `impulse.dump` the reference writes under -DDIAGNOSTIC for its Processing viewer
// (reference helpers.cpp:19-59, cmd/main.cpp:270-278, viewer/viewer.pde:55-70).  Declared unconditionally.
void print_diagnostic(unsigned long nrays, unsigned long nreflections, const std::vector<Impulse> & reflections,
                      const std::string & fname);

// Point on the unit sphere, -1 <= z <= 1, -pi <= theta <= pi (reference helpers.cpp:63-67).
cl_float3 spherePoint(float z, float theta);

// Uniformly random unit vectors, seeded from the wall clock like the reference (helpers.cpp:69-81).
std::vector<cl_float3> getRandomDirections(unsigned long num);

// Same distribution from a fixed seed (reproducible runs; not in the reference).
std::vector<cl_float3> getSeededDirections(unsigned long num, unsigned long seed);
