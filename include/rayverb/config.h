// config.h — JSON config validation layer, interface of reference rayverb/config.h: OutputMode,
// AttenuationModel and ConfigValidator with addRequiredValidator / addOptionalValidator / run.
// Same error behaviour: std::runtime_error("key <k> not found in config object") for a missing
// required key, std::runtime_error("invalid value") for a value of the wrong shape.
#pragma once

#include "rayverb.h"
#include "helpers.h"

#include "rapidjson/rapidjson.h"
#include "rapidjson/error/en.h"
#include "rapidjson/document.h"

#include <cmath>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

// which components of the impulse response are kept (reference config.h:19-23)
enum OutputMode
{   ALL
,   IMAGE_ONLY
,   DIFFUSE_ONLY
};

// reference config.h:29-38
struct AttenuationModel
{
    enum Mode
    {   SPEAKER
    ,   HRTF
    };
    Mode mode;
    HrtfConfig hrtf;
    std::vector<Speaker> speakers;
};

struct JsonValidatorBase
{
    virtual ~JsonValidatorBase() {}
    virtual void run(const rapidjson::Value & value) const = 0;
};

namespace rvbconfig {

inline void invalid() { throw std::runtime_error("invalid value"); }

// One overload per readable type: checks the shape (throws "invalid value") and stores the value.
inline void read(const rapidjson::Value & v, double & t) { if (!v.IsNumber()) invalid(); t = v.GetDouble(); }
inline void read(const rapidjson::Value & v, float & t) { if (!v.IsNumber()) invalid(); t = (float) v.GetDouble(); }
inline void read(const rapidjson::Value & v, bool & t) { if (!v.IsBool()) invalid(); t = v.GetBool(); }
inline void read(const rapidjson::Value & v, int & t) { if (!v.IsInt()) invalid(); t = v.GetInt(); }

template <typename T, int LENGTH>
inline void read_array(const rapidjson::Value & v, T & t)
{
    if (!v.IsArray() || v.Size() != LENGTH) invalid();
    for (int i = 0; i != LENGTH; ++i)
        if (!v[i].IsNumber()) invalid();
    for (int i = 0; i != LENGTH; ++i)
        t.s[i] = static_cast<cl_float>(v[i].GetDouble());
}
inline void read(const rapidjson::Value & v, cl_float3 & t) { read_array<cl_float3, 3>(v, t); }
inline void read(const rapidjson::Value & v, cl_float8 & t) { read_array<cl_float8, 8>(v, t); }

template <typename T>
inline void read_enum(const rapidjson::Value & v, T & t, const std::map<std::string, T> & names)
{
    if (!v.IsString()) invalid();
    auto it = names.find(v.GetString());
    if (it == names.end()) invalid();
    t = it->second;
}
inline void read(const rapidjson::Value & v, RayverbFiltering::FilterType & t)
{
    read_enum<RayverbFiltering::FilterType>(v, t, {{"sinc", RayverbFiltering::FILTER_TYPE_WINDOWED_SINC},
                                                   {"onepass", RayverbFiltering::FILTER_TYPE_BIQUAD_ONEPASS},
                                                   {"twopass", RayverbFiltering::FILTER_TYPE_BIQUAD_TWOPASS},
                                                   {"linkwitz_riley", RayverbFiltering::FILTER_TYPE_LINKWITZ_RILEY}});
}
inline void read(const rapidjson::Value & v, OutputMode & t)
{
    read_enum<OutputMode>(v, t, {{"all", ALL}, {"image_only", IMAGE_ONLY}, {"diffuse_only", DIFFUSE_ONLY}});
}

template <typename T> void read_required(const rapidjson::Value & v, const char * key, T & t);

inline void read(const rapidjson::Value & v, Surface & t)
{
    if (!v.IsObject()) invalid();
    read_required(v, "specular", t.specular);
    read_required(v, "diffuse", t.diffuse);
}
inline void read(const rapidjson::Value & v, Speaker & t)
{
    if (!v.IsObject()) invalid();
    read_required(v, "direction", t.direction);
    read_required(v, "shape", t.coefficient);
}
inline void normalize3(cl_float3 & v)        // reference config.h:396-405 (all four lanes are scaled)
{
    const cl_float len = 1.0 / std::sqrt(v.s[0] * v.s[0] + v.s[1] * v.s[1] + v.s[2] * v.s[2]);
    for (int i = 0; i != 4; ++i)
        v.s[i] *= len;
}
inline void read(const rapidjson::Value & v, HrtfConfig & t)
{
    if (!v.IsObject()) invalid();
    read_required(v, "facing", t.facing);
    read_required(v, "up", t.up);
    normalize3(t.facing);
    normalize3(t.up);
}
template <typename T>
inline void read(const rapidjson::Value & v, std::vector<T> & t)
{
    if (!v.IsArray()) invalid();
    for (auto i = v.Begin(); i != v.End(); ++i) {
        T temp = T();
        read(*i, temp);
        t.push_back(temp);
    }
}
inline void read(const rapidjson::Value & v, AttenuationModel & t)
{
    // exactly one of "speakers" / "hrtf" (reference config.h:445-456)
    if (!v.IsObject() || (v.HasMember("speakers") ? 1 : 0) + (v.HasMember("hrtf") ? 1 : 0) != 1) invalid();
    if (v.HasMember("speakers")) {
        t.mode = AttenuationModel::SPEAKER;
        read_required(v, "speakers", t.speakers);
    } else {
        t.mode = AttenuationModel::HRTF;
        read_required(v, "hrtf", t.hrtf);
    }
}

template <typename T>
inline void read_required(const rapidjson::Value & v, const char * key, T & t)
{
    if (!v.HasMember(key))
        throw std::runtime_error(std::string("key ") + key + " not found in config object");
    read(v[key], t);
}

}  // namespace rvbconfig

// Registers required and optional fields, then reads them all from a JSON object
// (reference config.h:58-81).
class ConfigValidator : public JsonValidatorBase
{
public:
    template <typename T>
    void addOptionalValidator(const std::string & s, T & t)
    {
        T * target = &t;
        validators.push_back([s, target](const rapidjson::Value & value) {
            if (value.HasMember(s.c_str()))
                rvbconfig::read(value[s.c_str()], *target);
        });
    }

    template <typename T>
    void addRequiredValidator(const std::string & s, T & t)
    {
        T * target = &t;
        validators.push_back([s, target](const rapidjson::Value & value) { rvbconfig::read_required(value, s.c_str(), *target); });
    }

    virtual void run(const rapidjson::Value & value) const
    {
        for (const auto & v : validators)
            v(value);
    }

private:
    std::vector<std::function<void(const rapidjson::Value &)>> validators;
};
