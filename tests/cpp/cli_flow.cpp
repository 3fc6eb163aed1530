// cli_flow.cpp — the flow of the reference's CLI (cmd/main.cpp:139-337) written against this
// repo's headers: JSON config -> Raytracer -> attenuator -> fixPredelay -> flattenImpulses ->
// process -> sound file.  The reference's own cmd/main.cpp compiles unchanged against the same
// headers (tests/test_cli_dropin.py checks that where /root/reference exists); this file is what
// runs on the GPU box.   cli_flow <config.json> <model.obj> <materials.json> <out.wav|aif>
#include "rayverb.h"
#include "helpers.h"
#include "config.h"

#include "rapidjson/document.h"
#include "rapidjson/error/en.h"
#include "sndfile.hh"

#include <iostream>
#include <map>

using namespace std;
using namespace rapidjson;

int main(int argc, char ** argv)
{
    if (argc != 5) { cerr << "usage: cli_flow config model materials output" << endl; return 1; }
    cl_float3 source = {{0, 0, 0, 0}}, mic = {{0, 0, 1, 0}};
    int numRays = 8192, numImpulses = 64, bitDepth = 16;
    double sampleRate = 44100.0, hipass = 45.0, volumme_scale = 1.0;
    auto filter = RayverbFiltering::FILTER_TYPE_BIQUAD_ONEPASS;
    bool normalize = true, trim_predelay = false, remove_direct = false, trim_tail = true, verbose = false;
    auto output_mode = ALL;
    AttenuationModel attenuationModel;

    Document document;
    attemptJsonParse(argv[1], document);
    if (document.HasParseError()) { cerr << GetParseError_En(document.GetParseError()) << endl; return 1; }
    if (!document.IsObject()) { cerr << "Rayverb config must be stored in a JSON object" << endl; return 1; }
    ConfigValidator cv;
    cv.addRequiredValidator("rays", numRays);
    cv.addRequiredValidator("reflections", numImpulses);
    cv.addRequiredValidator("sample_rate", sampleRate);
    cv.addRequiredValidator("bit_depth", bitDepth);
    cv.addRequiredValidator("source_position", source);
    cv.addRequiredValidator("mic_position", mic);
    cv.addRequiredValidator("attenuation_model", attenuationModel);
    cv.addOptionalValidator("filter", filter);
    cv.addOptionalValidator("hipass", hipass);
    cv.addOptionalValidator("normalize", normalize);
    cv.addOptionalValidator("volumme_scale", volumme_scale);
    cv.addOptionalValidator("trim_predelay", trim_predelay);
    cv.addOptionalValidator("remove_direct", remove_direct);
    cv.addOptionalValidator("trim_tail", trim_tail);
    cv.addOptionalValidator("output_mode", output_mode);
    cv.addOptionalValidator("verbose", verbose);
    try { cv.run(document); } catch (const runtime_error & e) { cerr << "config: " << e.what() << endl; return 1; }

    vector<vector<AttenuatedImpulse>> attenuated;
    try {
        auto directions = getSeededDirections(numRays, 1);
        Raytracer raytracer(numImpulses, argv[2], argv[3], verbose);
        raytracer.raytrace(mic, source, directions, verbose);
        RaytracerResults results = output_mode == ALL ? raytracer.getAllRaw(remove_direct)
                                 : output_mode == IMAGE_ONLY ? raytracer.getRawImages(remove_direct) : raytracer.getRawDiffuse();
        if (attenuationModel.mode == AttenuationModel::SPEAKER)
            attenuated = SpeakerAttenuator().attenuate(results, attenuationModel.speakers);
        else
            attenuated = HrtfAttenuator().attenuate(results, attenuationModel.hrtf.facing, attenuationModel.hrtf.up);
    } catch (const cl::Error & e) {
        cerr << "encountered opencl error:" << endl << e.what() << endl << e.err() << endl;
        return 2;
    } catch (const runtime_error & e) {
        cerr << "encountered runtime error:" << endl << e.what() << endl;
        return 3;
    }
    if (trim_predelay)
        fixPredelay(attenuated);
    auto flattened = flattenImpulses(attenuated, sampleRate);
    auto processed = process(filter, flattened, sampleRate, normalize, hipass, trim_tail, volumme_scale);

    vector<float> interleaved(processed.size() * processed[0].size());
    for (size_t i = 0; i != processed.size(); ++i)
        for (size_t j = 0; j != processed[i].size(); ++j)
            interleaved[j * processed.size() + i] = processed[i][j];
    const string out = argv[4];
    const bool wav = out.size() > 4 && out.substr(out.size() - 4) == ".wav";
    {
        SndfileHandle outfile(out, SFM_WRITE, (wav ? SF_FORMAT_WAV : SF_FORMAT_AIFF) | (bitDepth == 24 ? SF_FORMAT_PCM_24 : SF_FORMAT_PCM_16),
                              (int) processed.size(), (int) sampleRate);
        outfile.write(interleaved.data(), (sf_count_t) interleaved.size());
    }
    cout << "channels " << processed.size() << " frames " << processed[0].size() << endl;
    return 0;
}
