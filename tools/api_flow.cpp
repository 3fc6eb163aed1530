// api_flow.cpp — the reference's production call sequence (cmd/main.cpp:241-298) through the C++ mirror of its classes,
// timed stage by stage on a scene handed over as raw arrays:
//     Raytracer(nrefl, triangles, vertices, surfaces) -> raytrace -> getAllRaw -> SpeakerAttenuator().attenuate
//     -> fixPredelay -> flattenImpulses
// This is what a caller of the unchanged reference API gets (every stage's results cross the API as std::vector, as the
// reference's do).  bench.py runs it for the `api_flow_ms` leg; prints one JSON object.
//     api_flow <triangles.bin> <vertices.bin> <surfaces.bin> <directions.bin> <nrefl> sx sy sz mx my mz [repeats]
#include "rayverb/rayverb.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

template <class T> static std::vector<T> slurp(const char * path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "cannot open " << path << std::endl; std::exit(1); }
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t) bytes / sizeof(T));
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize) (v.size() * sizeof(T)));
    return v;
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char ** argv)
{
    if (argc < 12) { std::cerr << "usage: api_flow tris verts surfaces dirs nrefl sx sy sz mx my mz [repeats]" << std::endl; return 1; }
    std::vector<Triangle> triangles = slurp<Triangle>(argv[1]);
    std::vector<cl_float3> vertices = slurp<cl_float3>(argv[2]);
    std::vector<Surface> surfaces = slurp<Surface>(argv[3]);
    const std::vector<cl_float3> directions = slurp<cl_float3>(argv[4]);
    const unsigned long nrefl = std::strtoul(argv[5], nullptr, 10);
    const cl_float3 source = {{(float) atof(argv[6]), (float) atof(argv[7]), (float) atof(argv[8]), 0}};
    const cl_float3 mic = {{(float) atof(argv[9]), (float) atof(argv[10]), (float) atof(argv[11]), 0}};
    const int repeats = argc > 12 ? atoi(argv[12]) : 3;
    const std::vector<Speaker> speakers = {Speaker{{{-1, 0, -1, 0}}, 0.5f}, Speaker{{{1, 0, -1, 0}}, 0.5f}};
    try {
        double t0 = now_ms();
        Raytracer raytracer(nrefl, triangles, vertices, surfaces, false);
        const double scene_ms = now_ms() - t0;
        double best_total = 1e30, stage[5] = {0, 0, 0, 0, 0};
        size_t nbins = 0, nimpulses = 0;
        double checksum = 0;
        for (int r = 0; r < repeats; ++r) {
            double s[6];
            s[0] = now_ms();
            raytracer.raytrace(mic, source, directions, false);
            s[1] = now_ms();
            RaytracerResults results = raytracer.getAllRaw(false);
            s[2] = now_ms();
            std::vector<std::vector<AttenuatedImpulse>> attenuated = SpeakerAttenuator().attenuate(results, speakers);
            s[3] = now_ms();
            fixPredelay(attenuated);
            s[4] = now_ms();
            std::vector<std::vector<std::vector<float>>> flattened = flattenImpulses(attenuated, 44100.0f);
            s[5] = now_ms();
            if (s[5] - s[0] < best_total) {
                best_total = s[5] - s[0];
                for (int i = 0; i < 5; ++i) stage[i] = s[i + 1] - s[i];
            }
            nbins = flattened[0][0].size();
            nimpulses = results.impulses.size();
            checksum = 0;
            for (const auto & ch : flattened) for (const auto & band : ch) for (float v : band) checksum += v;
        }
        std::printf("{\"api_flow_ms\": %.3f, \"raytrace_ms\": %.3f, \"getAllRaw_ms\": %.3f, \"attenuate_ms\": %.3f, \"fixPredelay_ms\": %.3f, "
                    "\"flattenImpulses_ms\": %.3f, \"scene_ms\": %.3f, \"impulses\": %zu, \"nbins\": %zu, \"channels\": %zu, \"checksum\": %.9g, \"repeats\": %d}\n",
                    best_total, stage[0], stage[1], stage[2], stage[3], stage[4], scene_ms, nimpulses, nbins, speakers.size(), checksum, repeats);
    } catch (const cl::Error & e) {
        std::cerr << "cl::Error: " << e.what() << " (" << e.err() << ")" << std::endl;
        return 2;
    }
    return 0;
}
