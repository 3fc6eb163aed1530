// postprocess.cpp — the crossover filters and the post-processing chain (reference rayverb/filters.cpp,
// rayverb/rayverb.cpp:79-149).  Host-side O(samples) work on a few hundred thousand samples.
#include "../../include/rayverb/rayverb.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <mutex>
#include <thread>
#include <iterator>
#include <memory>
#include <stdexcept>

using std::vector;

namespace {

const unsigned long KERNEL_LENGTH = 29;       // reference filters.h:124, :139

// sin(pi t) / (pi t)
double sinc(double t)
{
    const double pit = M_PI * t;
    return std::sin(pit) / pit;
}

// un-windowed low-pass sinc taps (reference filters.cpp:17-34); cutoff as a fraction of the sample rate
vector<float> sincKernel(double cutoff, unsigned long length)
{
    if (!(length % 2))
        throw std::runtime_error("Length of sinc filter kernel must be odd.");
    vector<float> ret(length);
    for (unsigned long i = 0; i != length; ++i)
        ret[i] = i == (length - 1) / 2 ? 1.0f : (float) sinc(2 * cutoff * ((double) i - (length - 1) / 2.0));
    return ret;
}

vector<float> blackman(unsigned long length)
{
    const double a0 = 7938.0 / 18608.0, a1 = 9240.0 / 18608.0, a2 = 1430.0 / 18608.0;
    vector<float> ret(length);
    for (unsigned long i = 0; i != length; ++i) {
        const double offset = i / (length - 1.0);
        ret[i] = (float) (a0 - a1 * std::cos(2 * M_PI * offset) + a2 * std::cos(4 * M_PI * offset));
    }
    return ret;
}

// windowed, peak-normalised low-pass taps (reference filters.cpp:57-71)
vector<float> lopassKernel(float sr, float cutoff, unsigned long length)
{
    vector<float> window = blackman(length);
    vector<float> kernel = sincKernel(cutoff / sr, length);
    for (unsigned long i = 0; i != length; ++i)
        kernel[i] = window[i] * kernel[i];
    normalize(kernel);
    return kernel;
}

// spectral inversion of the low-pass (reference filters.cpp:75-81)
vector<float> hipassKernel(float sr, float cutoff, unsigned long length)
{
    vector<float> kernel = lopassKernel(sr, cutoff, length);
    for (float & i : kernel) i = -i;
    kernel[(length - 1) / 2] += 1;
    return kernel;
}

// Full linear convolution, zero-padded to `out_length`, multiplied by `out_length`: what the
// reference's FastConvolution::convolve returns, because FFTW's c2r transform is unnormalised
// (reference filters.h:56-80).
vector<float> convolveScaled(const vector<float> & a, const vector<float> & b, unsigned long out_length)
{
    vector<double> acc(out_length, 0.0);
    for (size_t i = 0; i < a.size(); ++i) {
        if (a[i] == 0.0f) continue;
        for (size_t j = 0; j < b.size() && i + j < out_length; ++j)
            acc[i + j] += (double) a[i] * b[j];
    }
    vector<float> out(out_length);
    for (unsigned long i = 0; i < out_length; ++i)
        out[i] = (float) (acc[i] * (double) out_length);
    return out;
}

struct Bandpass {
    virtual ~Bandpass() {}
    virtual void setParams(float l, float h, float s) = 0;
    virtual void filter(vector<float> & data) = 0;
};

// reference filters.cpp:118-154
struct BandpassWindowedSinc : Bandpass {
    vector<float> kernel;
    void setParams(float l, float h, float s)
    {
        vector<float> lop = lopassKernel(s, h, 1 + KERNEL_LENGTH / 2);
        vector<float> hip = hipassKernel(s, l, 1 + KERNEL_LENGTH / 2);
        kernel = convolveScaled(lop, hip, KERNEL_LENGTH);
    }
    void filter(vector<float> & data) { data = convolveScaled(kernel, data, KERNEL_LENGTH + data.size() - 1); }
};

// Second-order sections as {b0, b1, b2, a1, a2}, already divided by a0.
struct Section { double b0, b1, b2, a1, a2; };

// Constant-skirt band-pass of the Audio-EQ cookbook between two corner frequencies (what the reference designs at filters.cpp:198-223):
// centre = geometric mean of the corners, width in octaves, Q from the width at the centre's digital frequency.
Section bandpassSection(float corner_lo, float corner_hi, float sample_rate)
{
    const double centre = std::sqrt(corner_lo * corner_hi);
    const double w0 = 2 * M_PI * centre / sample_rate;
    const double cos_w0 = std::cos(w0), sin_w0 = std::sin(w0);
    const double octaves = std::log2(corner_hi / corner_lo);
    const double quality = sin_w0 / (std::log(2) * octaves * w0);
    const double half_width = sin_w0 * std::sinh(1 / (2 * quality));
    const double gain = 1 / (1 + half_width);                      // 1 / a0
    return Section{gain * half_width, gain * 0, gain * -half_width, gain * (-2 * cos_w0), gain * (1 - half_width)};
}

// Second-order Butterworth low-pass / high-pass by the bilinear transform with k = cot(pi fc / fs) (reference filters.cpp:225-266)
struct ButterworthPair { Section low, high; };
double cotangentOfHalfDigitalFrequency(double cutoff, double sample_rate)
{
    const double half = M_PI * cutoff / sample_rate;
    return std::cos(half) / std::sin(half);
}
Section butterworth(double cutoff, double sample_rate, bool highpass)
{
    const double k = cotangentOfHalfDigitalFrequency(cutoff, sample_rate);
    const double k2 = k * k, damping = k * std::sqrt(2);
    const double a0 = k2 + damping + 1;
    const double a1 = (-2 * (k2 - 1)) / a0, a2 = (k2 - damping + 1) / a0;
    return highpass ? Section{k2 / a0, (-2 * k2) / a0, k2 / a0, a1, a2} : Section{1 / a0, 2 / a0, 1 / a0, a1, a2};
}

void load(RayverbFiltering::Biquad & filter, const Section & s) { filter.setParams(s.b0, s.b1, s.b2, s.a1, s.a2); }

struct OnepassBandpassBiquad : Bandpass, RayverbFiltering::Biquad {
    void setParams(float lo, float hi, float sr) { load(*this, bandpassSection(lo, hi, sr)); }
    void filter(vector<float> & data) { onepass(data); }
};

struct TwopassBandpassBiquad : OnepassBandpassBiquad {
    void filter(vector<float> & data) { twopass(data); }
};

// zero-phase band-pass from a low-pass at the upper corner and a high-pass at the lower one, each run forwards and backwards
struct LinkwitzRiley : Bandpass {
    RayverbFiltering::Biquad lopass, hipass;
    void setParams(float lower, float upper, float sample_rate)
    {
        load(lopass, butterworth(upper, sample_rate, false));
        load(hipass, butterworth(lower, sample_rate, true));
    }
    void filter(vector<float> & data)
    {
        lopass.twopass(data);
        hipass.twopass(data);
    }
};

}  // namespace

void RayverbFiltering::Biquad::setParams(double _b0, double _b1, double _b2, double _a1, double _a2)
{
    b0 = _b0; b1 = _b1; b2 = _b2; a1 = _a1; a2 = _a2;
}

void RayverbFiltering::Biquad::onepass(vector<float> & data)
{
    double z1 = 0, z2 = 0;
    for (float & i : data) {
        const double out = i * b0 + z1;
        z1 = i * b1 + z2 - a1 * out;
        z2 = i * b2 - a2 * out;
        i = (float) out;
    }
}

void RayverbFiltering::Biquad::twopass(vector<float> & data)
{
    onepass(data);
    std::reverse(data.begin(), data.end());
    onepass(data);
    std::reverse(data.begin(), data.end());
}

namespace {
Bandpass * makeBandpass(RayverbFiltering::FilterType ft)
{
    switch (ft) {
    case RayverbFiltering::FILTER_TYPE_WINDOWED_SINC: return new BandpassWindowedSinc();
    case RayverbFiltering::FILTER_TYPE_BIQUAD_ONEPASS: return new OnepassBandpassBiquad();
    case RayverbFiltering::FILTER_TYPE_BIQUAD_TWOPASS: return new TwopassBandpassBiquad();
    case RayverbFiltering::FILTER_TYPE_LINKWITZ_RILEY: return new LinkwitzRiley();
    }
    return nullptr;
}
}  // namespace

// The reference filters the (channel, band) signals one after the other (filters.cpp:268-306).  They are independent — every one is a
// serial recurrence over its own samples — so here each runs on a host thread of its own (at workload C2, 2 x 8 signals of 846 741
// samples: 0.08-0.55 s on one core -> 16 threads); every signal's arithmetic is the serial one, so the result does not depend on the
// thread count (RVB_FILTER_THREADS=1 for the serial order).
void RayverbFiltering::filter(FilterType ft, vector<vector<vector<float>>> & data, float sr, float lo_cutoff)
{
    const float EDGES[9] = {lo_cutoff, 175, 350, 700, 1400, 2800, 5600, 11200, 20000};
    struct Task { vector<float> * signal; float lo, hi; };
    vector<Task> tasks;
    for (auto & channel : data)
        for (size_t i = 0; i != channel.size() && i < 8; ++i)
            tasks.push_back(Task{&channel[i], EDGES[i], EDGES[i + 1]});
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char * env = std::getenv("RVB_FILTER_THREADS")) nthreads = (unsigned) std::max(1, std::atoi(env));
    nthreads = std::max(1u, std::min<unsigned>(nthreads, (unsigned) tasks.size()));
    std::atomic<size_t> next(0);
    std::exception_ptr failure;
    std::mutex failure_lock;
    auto work = [&]() {
        try {
            std::unique_ptr<Bandpass> bp(makeBandpass(ft));
            for (size_t t = next++; t < tasks.size(); t = next++) {
                bp->setParams(tasks[t].lo, tasks[t].hi, sr);
                bp->filter(*tasks[t].signal);
            }
        } catch (...) {
            std::lock_guard<std::mutex> guard(failure_lock);
            if (!failure) failure = std::current_exception();
        }
    };
    if (nthreads == 1) {
        work();
    } else {
        vector<std::thread> pool;
        for (unsigned k = 0; k < nthreads; ++k) pool.emplace_back(work);
        for (std::thread & t : pool) t.join();
    }
    if (failure) std::rethrow_exception(failure);
}

// ---- mixdown / trim / process (reference rayverb.cpp:79-149) -------------------------------------------

namespace {

vector<float> mixdown(const vector<vector<float>> & data)
{
    vector<float> ret(data.front().size(), 0);
    for (const auto & band : data)
        for (size_t i = 0; i < ret.size() && i < band.size(); ++i)
            ret[i] = ret[i] + band[i];
    return ret;
}

// Keeps samples up to, NOT including, the last one whose magnitude reaches minVol
// (the reference drops that sample too — quirk Q8, rayverb.cpp:96-122).
void trimTail(vector<vector<float>> & audioChannels, float minVol)
{
    long len = 0;
    for (const auto & ch : audioChannels) {
        long last = -1;
        for (long i = (long) ch.size() - 1; i >= 0; --i)
            if (std::fabs(ch[(size_t) i]) >= minVol) { last = i; break; }
        len = std::max(len, last);       // distance(begin, found.base()) - 1
    }
    for (auto & ch : audioChannels)
        ch.resize((size_t) len);
}

}  // namespace

vector<vector<float>> process(RayverbFiltering::FilterType filtertype, vector<vector<vector<float>>> & data, float sr,
                              bool do_normalize, float lo_cutoff, bool do_trim_tail, float volume_scale)
{
    RayverbFiltering::filter(filtertype, data, sr, lo_cutoff);
    vector<vector<float>> ret(data.size());
    for (size_t i = 0; i < data.size(); ++i)
        ret[i] = mixdown(data[i]);
    if (do_normalize)
        normalize(ret);
    if (volume_scale != 1)
        mul(ret, volume_scale);
    if (do_trim_tail)
        trimTail(ret, 0.00001);
    return ret;
}

void attemptJsonParse(const std::string & fname, rapidjson::Document & doc)
{
    std::ifstream in(fname);
    std::string file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    doc.Parse(file.c_str());
}
