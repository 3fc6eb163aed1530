// postprocess_tool.cpp — test driver: reads [channels][8][n] float32 from a file, runs
// process() (reference rayverb.cpp:125-149) with the given options, writes [channels][m] float32.
//   postprocess_tool in.bin out.bin channels n filter(0..3) sr normalize lo_cutoff trim_tail volume_scale
#include "rayverb.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char ** argv)
{
    if (argc != 11) return 64;
    const int channels = std::atoi(argv[3]);
    const long n = std::atol(argv[4]);
    std::vector<std::vector<std::vector<float>>> data(channels, std::vector<std::vector<float>>(8, std::vector<float>(n)));
    FILE * f = std::fopen(argv[1], "rb");
    if (!f) return 65;
    for (auto & ch : data)
        for (auto & band : ch)
            if (std::fread(band.data(), sizeof(float), n, f) != (size_t) n) return 66;
    std::fclose(f);
    std::vector<std::vector<float>> out = process((RayverbFiltering::FilterType) std::atoi(argv[5]), data, (float) std::atof(argv[6]),
                                                  std::atoi(argv[7]) != 0, (float) std::atof(argv[8]), std::atoi(argv[9]) != 0,
                                                  (float) std::atof(argv[10]));
    f = std::fopen(argv[2], "wb");
    if (!f) return 67;
    for (auto & ch : out) {
        const long m = (long) ch.size();
        std::fwrite(&m, sizeof(m), 1, f);
        std::fwrite(ch.data(), sizeof(float), ch.size(), f);
    }
    std::fclose(f);
    return 0;
}
