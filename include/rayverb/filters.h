// filters.h — the 8-band crossover applied to the binned impulse response before mixdown,
// interface of reference rayverb/filters.h (namespace RayverbFiltering).  Host-side O(samples)
// post-processing, not part of the GPU hot path.  No FFTW: the windowed-sinc
// mode convolves its 29-tap kernels directly (and reproduces the reference's unnormalised
// FFTW scaling so that un-normalised output has the same gain).
#pragma once

#include <vector>

namespace RayverbFiltering
{
    // reference filters.h:196-201
    enum FilterType
    {   FILTER_TYPE_WINDOWED_SINC
    ,   FILTER_TYPE_BIQUAD_ONEPASS
    ,   FILTER_TYPE_BIQUAD_TWOPASS
    ,   FILTER_TYPE_LINKWITZ_RILEY
    };

    // Direct-form-II-transposed biquad with double state (reference filters.cpp:156-196)
    class Biquad
    {
    public:
        void onepass(std::vector<float> & data);
        void twopass(std::vector<float> & data);      // forward, then backward: zero phase
        void setParams(double b0, double b1, double b2, double a1, double a2);
    private:
        double b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0;
    };

    // Band-pass every band of every channel in place: data[channel][band][sample]; band edges
    // {lo_cutoff, 175, 350, 700, 1400, 2800, 5600, 11200, 20000} Hz (reference filters.cpp:268-306).
    void filter(FilterType ft, std::vector<std::vector<std::vector<float>>> & data, float sr, float lo_cutoff);
}
