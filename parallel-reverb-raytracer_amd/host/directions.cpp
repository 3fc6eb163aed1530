// directions.cpp — ray directions (reference rayverb/helpers.cpp:63-81).
#include "../../include/rayverb/helpers.h"

#include <chrono>
#include <cmath>
#include <random>

cl_float3 spherePoint(float z, float theta)
{
    const float ztemp = sqrtf(1 - z * z);
    return cl_float3{{ztemp * cosf(theta), ztemp * sinf(theta), z, 0}};
}

std::vector<cl_float3> getSeededDirections(unsigned long num, unsigned long seed)
{
    std::vector<cl_float3> ret(num);
    std::uniform_real_distribution<float> zDist(-1, 1);
    std::uniform_real_distribution<float> thetaDist((float) -M_PI, (float) M_PI);
    std::default_random_engine engine(seed);
    for (cl_float3 & i : ret)
        i = spherePoint(zDist(engine), thetaDist(engine));
    return ret;
}

std::vector<cl_float3> getRandomDirections(unsigned long num)
{
    return getSeededDirections(num, (unsigned long) std::chrono::system_clock::now().time_since_epoch().count());
}
