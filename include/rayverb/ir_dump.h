// ir_dump.h — compact binary dump of an impulse response as the hot path produces it: the per-channel,
// per-band time histograms [channels][8][nbins] plus the merged image-source impulses.
// The reference only keeps such data in memory between flattenImpulses (rayverb.cpp:28-77) and the
// filter chain; the file form exists for offline comparison of runs and for shipping a rank's / node's
// partial histogram elsewhere.  Little-endian, no padding:
//
//   char     magic[8]  = "RVBHIST1"
//   uint32   channels, bands (= 8)
//   uint64   nbins
//   float32  sample_rate, predelay_seconds
//   uint64   nimages
//   float32  histogram[channels][bands][nbins]
//   Impulse  images[nimages]                    (64-byte records of clstructs.h)
#pragma once

#include "clstructs.h"

#include <string>
#include <vector>

struct IrDump {
    unsigned channels = 0;
    unsigned long nbins = 0;
    float sample_rate = 0, predelay = 0;
    std::vector<float> histogram;        // [channels][8][nbins]
    std::vector<Impulse> images;
};

// Throws std::runtime_error on I/O failure or a malformed file.
void write_ir_dump(const std::string & fname, const IrDump & dump);
IrDump read_ir_dump(const std::string & fname);
