// sndfile.hh — the sliver of libsndfile's C++ wrapper that the reference's CLI uses
// (cmd/main.cpp:26-48, :209-239): SndfileHandle(path, SFM_WRITE, format, channels, samplerate) and
// write(const float *, count) for 16/24-bit PCM WAV and AIFF.  libsndfile is not available offline;
// this is an independent header-only writer.  Samples are scaled by 2^(bits-1)-1, rounded and
// clipped; the file is written when the handle is destroyed.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

typedef long long sf_count_t;

enum {
    SFM_READ = 0x10, SFM_WRITE = 0x20,
    SF_FORMAT_WAV = 0x010000, SF_FORMAT_AIFF = 0x020000,
    SF_FORMAT_PCM_16 = 0x0002, SF_FORMAT_PCM_24 = 0x0003,
    SF_FORMAT_SUBMASK = 0x0000FFFF, SF_FORMAT_TYPEMASK = 0x0FFF0000
};

class SndfileHandle {
public:
    SndfileHandle(const std::string & path, int mode = SFM_READ, int format = 0, int channels = 0, int samplerate = 0)
        : path_(path), mode_(mode), format_(format), channels_(channels), samplerate_(samplerate) {}
    ~SndfileHandle() { if (mode_ == SFM_WRITE) flush(); }

    sf_count_t write(const float * ptr, sf_count_t items)
    {
        samples_.insert(samples_.end(), ptr, ptr + items);
        return items;
    }
    int channels() const { return channels_; }
    int samplerate() const { return samplerate_; }
    int format() const { return format_; }

private:
    std::string path_;
    int mode_, format_, channels_, samplerate_;
    std::vector<float> samples_;

    SndfileHandle(const SndfileHandle &);
    SndfileHandle & operator=(const SndfileHandle &);

    static void be32(std::vector<unsigned char> & o, uint32_t v) { for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char) (v >> s)); }
    static void be16(std::vector<unsigned char> & o, uint32_t v) { o.push_back((unsigned char) (v >> 8)); o.push_back((unsigned char) v); }
    static void le32(std::vector<unsigned char> & o, uint32_t v) { for (int s = 0; s < 32; s += 8) o.push_back((unsigned char) (v >> s)); }
    static void le16(std::vector<unsigned char> & o, uint32_t v) { o.push_back((unsigned char) v); o.push_back((unsigned char) (v >> 8)); }
    static void tag(std::vector<unsigned char> & o, const char * t) { o.insert(o.end(), t, t + 4); }

    void flush()
    {
        const int bits = (format_ & SF_FORMAT_SUBMASK) == SF_FORMAT_PCM_24 ? 24 : 16;
        const int bytes = bits / 8;
        const bool aiff = (format_ & SF_FORMAT_TYPEMASK) == SF_FORMAT_AIFF;
        const double scale = bits == 24 ? 8388607.0 : 32767.0;
        std::vector<unsigned char> pcm;
        pcm.reserve(samples_.size() * (size_t) bytes);
        for (float s : samples_) {
            double v = std::floor((double) s * scale + 0.5);
            if (v > scale) v = scale;
            if (v < -scale - 1) v = -scale - 1;
            const int32_t q = (int32_t) v;
            if (aiff) for (int b = bytes - 1; b >= 0; --b) pcm.push_back((unsigned char) (q >> (8 * b)));
            else for (int b = 0; b < bytes; ++b) pcm.push_back((unsigned char) (q >> (8 * b)));
        }
        const uint32_t frames = channels_ > 0 ? (uint32_t) (samples_.size() / (size_t) channels_) : 0;
        std::vector<unsigned char> out;
        if (aiff) {
            tag(out, "FORM"); be32(out, (uint32_t) (4 + 26 + 16 + pcm.size())); tag(out, "AIFF");
            tag(out, "COMM"); be32(out, 18); be16(out, (uint32_t) channels_); be32(out, frames); be16(out, (uint32_t) bits);
            // 80-bit IEEE extended sample rate
            int exponent = 0;
            double mant = std::frexp((double) samplerate_, &exponent);   // rate = mant * 2^exponent, mant in [0.5, 1)
            const uint64_t m = samplerate_ > 0 ? (uint64_t) std::ldexp(mant, 64) : 0;
            be16(out, samplerate_ > 0 ? (uint32_t) (16383 + exponent - 1) : 0);
            be32(out, (uint32_t) (m >> 32)); be32(out, (uint32_t) m);
            tag(out, "SSND"); be32(out, (uint32_t) (8 + pcm.size())); be32(out, 0); be32(out, 0);
        } else {
            tag(out, "RIFF"); le32(out, (uint32_t) (36 + pcm.size())); tag(out, "WAVE");
            tag(out, "fmt "); le32(out, 16); le16(out, 1); le16(out, (uint32_t) channels_); le32(out, (uint32_t) samplerate_);
            le32(out, (uint32_t) (samplerate_ * channels_ * bytes)); le16(out, (uint32_t) (channels_ * bytes)); le16(out, (uint32_t) bits);
            tag(out, "data"); le32(out, (uint32_t) pcm.size());
        }
        out.insert(out.end(), pcm.begin(), pcm.end());
        if (FILE * f = std::fopen(path_.c_str(), "wb")) {
            std::fwrite(out.data(), 1, out.size(), f);
            std::fclose(f);
        }
    }
};
