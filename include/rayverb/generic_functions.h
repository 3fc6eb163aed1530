// generic_functions.h — small recursive helpers over nested vectors, interface of reference
// rayverb/generic_functions.h (max_amp, div, mul, normalize, elementwise).
#pragma once

#include <algorithm>
#include <cmath>
#include <iterator>
#include <vector>

inline float max_amp(const float & t) { return std::fabs(t); }
template <typename T>
inline float max_amp(const std::vector<T> & t)
{
    float m = 0.0f;
    for (const T & i : t)
        m = std::max(m, max_amp(i));
    return m;
}

inline void div(float & ret, float f) { ret /= f; }
template <typename T>
inline void div(std::vector<T> & ret, float f)
{
    for (T & i : ret)
        div(i, f);
}

inline void mul(float & ret, float f) { ret *= f; }
template <typename T>
inline void mul(std::vector<T> & ret, float f)
{
    for (T & i : ret)
        mul(i, f);
}

// scale so that the largest magnitude becomes 1 (reference generic_functions.h:57-63)
template <typename T>
inline void normalize(std::vector<T> & ret)
{
    mul(ret, 1.0 / max_amp(ret));
}

// binary operation over the lanes of two cl_floatN values
template <typename T, typename U>
inline T elementwise(const T & a, const T & b, const U & u)
{
    T ret;
    std::transform(std::begin(a.s), std::end(a.s), std::begin(b.s), std::begin(ret.s), u);
    return ret;
}
