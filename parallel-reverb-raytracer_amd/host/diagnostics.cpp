// diagnostics.cpp — on-disk formats around the hot path: the reference's `impulse.dump`
// (reference rayverb/helpers.cpp:19-59) and the binary impulse-response dump of include/rayverb/ir_dump.h.
#include "../../include/rayverb/helpers.h"
#include "../../include/rayverb/ir_dump.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace {

// A double as rapidjson's Writer::Double prints it (the reference's writer): the shortest decimal digits that
// read back as the same double, laid out by rapidjson's Prettify rules — plain decimals with at least ".0"
// for decimal exponents in (-6, 21], otherwise d.ddde[-]x.  rapidjson finds the digits with Grisu2, which in
// rare cases (<0.1 %) emits one digit more than the shortest; that choice of digits is parity-unpinned
// (rapidjson is not available here), the value read back is identical either way.
std::string json_double(double d)
{
    if (std::isnan(d) || std::isinf(d))
        return "null";                                     // rapidjson refuses NaN/Inf; a reader sees null
    if (d == 0.0)
        return std::signbit(d) ? "-0.0" : "0.0";
    char buf[40];
    int prec = 1;
    for (; prec <= 17; ++prec) {
        std::snprintf(buf, sizeof(buf), "%.*e", prec - 1, d);
        if (std::strtod(buf, nullptr) == d)
            break;
    }
    // buf = [-]d.ddddde[+-]xx
    std::string s(buf);
    std::string out;
    size_t pos = 0;
    if (s[0] == '-') { out = "-"; pos = 1; }
    const size_t e = s.find('e');
    std::string digits;
    for (size_t i = pos; i < e; ++i)
        if (s[i] != '.') digits += s[i];
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int exp10 = std::atoi(s.c_str() + e + 1);
    const int length = (int) digits.size();
    const int k = exp10 - (length - 1);                    // value = digits * 10^k
    const int kk = length + k;                             // position of the decimal point
    if (0 <= k && kk <= 21) {
        out += digits + std::string((size_t) k, '0') + ".0";
    } else if (0 < kk && kk <= 21) {
        out += digits.substr(0, (size_t) kk) + "." + digits.substr((size_t) kk);
    } else if (-6 < kk && kk <= 0) {
        out += "0." + std::string((size_t) -kk, '0') + digits;
    } else {
        out += digits.substr(0, 1);
        if (length > 1) out += "." + digits.substr(1);
        out += "e" + std::to_string(kk - 1);
    }
    return out;
}

}  // namespace

void print_diagnostic(unsigned long nrays, unsigned long nreflections, const std::vector<Impulse> & impulses,
                      const std::string & fname)
{
    if (impulses.size() < (size_t) nrays * nreflections)
        throw std::runtime_error("print_diagnostic: fewer impulses than nrays * nreflections");
    std::ofstream out(fname);
    if (!out)
        throw std::runtime_error("print_diagnostic: cannot open " + fname);
    std::string line;
    for (unsigned long i = 0; i != nrays; ++i) {
        line = "[";
        for (unsigned long j = 0; j != nreflections; ++j) {
            const Impulse & r = impulses[i * nreflections + j];
            if (j) line += ",";
            line += "{\"position\":[";
            for (int k = 0; k != 3; ++k) {
                if (k) line += ",";
                line += json_double(r.position.s[k]);
            }
            line += "],\"volume\":";
            float average = 0;                             // float accumulation in band order, helpers.cpp:46-49
            for (int k = 0; k != 8; ++k)
                average += r.volume.s[k];
            average /= 8;
            line += json_double(average) + "}";
        }
        line += "]";
        out << line << std::endl;
    }
}

void write_ir_dump(const std::string & fname, const IrDump & dump)
{
    if (dump.histogram.size() != (size_t) dump.channels * 8 * dump.nbins)
        throw std::runtime_error("write_ir_dump: histogram size does not match channels * 8 * nbins");
    std::ofstream out(fname, std::ios::binary);
    if (!out)
        throw std::runtime_error("write_ir_dump: cannot open " + fname);
    const uint32_t channels = dump.channels, bands = 8;
    const uint64_t nbins = dump.nbins, nimages = dump.images.size();
    out.write("RVBHIST1", 8);
    out.write(reinterpret_cast<const char *>(&channels), 4);
    out.write(reinterpret_cast<const char *>(&bands), 4);
    out.write(reinterpret_cast<const char *>(&nbins), 8);
    out.write(reinterpret_cast<const char *>(&dump.sample_rate), 4);
    out.write(reinterpret_cast<const char *>(&dump.predelay), 4);
    out.write(reinterpret_cast<const char *>(&nimages), 8);
    out.write(reinterpret_cast<const char *>(dump.histogram.data()), (std::streamsize) (dump.histogram.size() * sizeof(float)));
    out.write(reinterpret_cast<const char *>(dump.images.data()), (std::streamsize) (dump.images.size() * sizeof(Impulse)));
    if (!out)
        throw std::runtime_error("write_ir_dump: write failed on " + fname);
}

IrDump read_ir_dump(const std::string & fname)
{
    std::ifstream in(fname, std::ios::binary);
    if (!in)
        throw std::runtime_error("read_ir_dump: cannot open " + fname);
    char magic[8];
    uint32_t channels = 0, bands = 0;
    uint64_t nbins = 0, nimages = 0;
    IrDump d;
    in.read(magic, 8);
    in.read(reinterpret_cast<char *>(&channels), 4);
    in.read(reinterpret_cast<char *>(&bands), 4);
    in.read(reinterpret_cast<char *>(&nbins), 8);
    in.read(reinterpret_cast<char *>(&d.sample_rate), 4);
    in.read(reinterpret_cast<char *>(&d.predelay), 4);
    in.read(reinterpret_cast<char *>(&nimages), 8);
    if (!in || std::memcmp(magic, "RVBHIST1", 8) != 0 || bands != 8)
        throw std::runtime_error("read_ir_dump: " + fname + " is not an RVBHIST1 file");
    if (channels > 64 || nbins > (1ull << 40) || nimages > (1ull << 32))
        throw std::runtime_error("read_ir_dump: implausible header in " + fname);
    d.channels = channels;
    d.nbins = nbins;
    d.histogram.resize((size_t) channels * 8 * nbins);
    d.images.resize(nimages);
    in.read(reinterpret_cast<char *>(d.histogram.data()), (std::streamsize) (d.histogram.size() * sizeof(float)));
    in.read(reinterpret_cast<char *>(d.images.data()), (std::streamsize) (d.images.size() * sizeof(Impulse)));
    if (!in)
        throw std::runtime_error("read_ir_dump: truncated file " + fname);
    return d;
}
