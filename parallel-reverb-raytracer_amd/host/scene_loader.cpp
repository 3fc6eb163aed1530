// scene_loader.cpp — see scene_loader.h.
//
// What is kept from the reference's behaviour (rayverb/rayverb.cpp:331-445):
//   * surfaces[0] is the built-in default surface; the material file's entries follow in
//     std::map (bytewise name) order; a mesh whose material name is not in the file gets surface 0;
//   * polygons are triangulated, vertices are taken as they are in the file.
// What cannot be reproduced: Assimp's exact triangle order and its per-mesh vertex duplication
// (third-party, SURVEY.md §8(c) gap 2).  Polygons are ear-clipped here (concave polygons occur in
// the reference's demo models, e.g. bedroom.obj); results depend on the triangle SET only up to
// the tie rule "lowest triangle index wins", so positions/times/volumes are unaffected.
#include "scene_loader.h"
#include "rvb_json.h"

#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace {

std::string slurp(const std::string & fname)
{
    std::ifstream in(fname);
    if (!in)
        throw std::runtime_error("Failed to open file: " + fname);
    std::stringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

VolumeType readBands(const rvbjson::Value & entry, const char * key, const std::string & name)
{
    const rvbjson::Value * a = entry.find(key);
    if (!a || !a->isArray() || a->array.size() != 8)
        throw std::runtime_error("material '" + name + "': '" + key + "' must be an array of 8 numbers");
    VolumeType v;
    for (int i = 0; i < 8; ++i) {
        if (!a->array[i].isNumber())
            throw std::runtime_error("material '" + name + "': '" + key + "' must be an array of 8 numbers");
        v.s[i] = (float) a->array[i].number;
    }
    return v;
}

struct P3 { double x, y, z; };

// Ear clipping of one polygon (indices into `pts`), robust enough for planar SketchUp exports:
// project on the plane of largest normal component (Newell normal), clip convex ears that contain
// no other vertex; fall back to a fan if no ear is found (degenerate input).
void triangulate(const std::vector<cl_float3> & verts, const std::vector<cl_ulong> & poly, cl_ulong surface,
                 std::vector<Triangle> & out)
{
    const size_t n = poly.size();
    if (n < 3)
        return;
    if (n == 3) {
        out.push_back(Triangle{surface, poly[0], poly[1], poly[2]});
        return;
    }
    double nx = 0, ny = 0, nz = 0;
    for (size_t i = 0; i < n; ++i) {
        const cl_float3 & a = verts[poly[i]];
        const cl_float3 & b = verts[poly[(i + 1) % n]];
        nx += ((double) a.s[1] - b.s[1]) * ((double) a.s[2] + b.s[2]);
        ny += ((double) a.s[2] - b.s[2]) * ((double) a.s[0] + b.s[0]);
        nz += ((double) a.s[0] - b.s[0]) * ((double) a.s[1] + b.s[1]);
    }
    int drop = 0;
    if (std::fabs(ny) > std::fabs(nx)) drop = 1;
    if (std::fabs(nz) > std::fabs(drop == 0 ? nx : ny)) drop = 2;
    const int ax = (drop + 1) % 3, ay = (drop + 2) % 3;
    const double orient = (drop == 0 ? nx : drop == 1 ? ny : nz) >= 0 ? 1.0 : -1.0;
    auto px = [&](cl_ulong v) { return (double) verts[v].s[ax]; };
    auto py = [&](cl_ulong v) { return (double) verts[v].s[ay]; };
    auto cross = [&](cl_ulong a, cl_ulong b, cl_ulong c) {
        return orient * ((px(b) - px(a)) * (py(c) - py(a)) - (py(b) - py(a)) * (px(c) - px(a)));
    };
    std::vector<cl_ulong> ring(poly);
    while (ring.size() > 3) {
        const size_t m = ring.size();
        bool clipped = false;
        for (size_t i = 0; i < m && !clipped; ++i) {
            const cl_ulong a = ring[(i + m - 1) % m], b = ring[i], c = ring[(i + 1) % m];
            if (cross(a, b, c) <= 0)
                continue;                                   // reflex or collinear corner
            bool empty = true;
            for (size_t k = 0; k < m && empty; ++k) {
                const cl_ulong p = ring[k];
                if (p == a || p == b || p == c)
                    continue;
                if (cross(a, b, p) >= 0 && cross(b, c, p) >= 0 && cross(c, a, p) >= 0)
                    empty = false;
            }
            if (!empty)
                continue;
            out.push_back(Triangle{surface, a, b, c});
            ring.erase(ring.begin() + (long) i);
            clipped = true;
        }
        if (!clipped) {                                     // degenerate: fan the rest
            for (size_t k = 1; k + 1 < ring.size(); ++k)
                out.push_back(Triangle{surface, ring[0], ring[k], ring[k + 1]});
            return;
        }
    }
    out.push_back(Triangle{surface, ring[0], ring[1], ring[2]});
}

}  // namespace

std::map<std::string, Surface> loadMaterials(const std::string & materialFileName)
{
    const rvbjson::Value doc = rvbjson::parse(slurp(materialFileName));
    if (!doc.isObject())
        throw std::runtime_error("Materials must be stored in a JSON object");     // rayverb.cpp:308-309
    std::map<std::string, Surface> ret;
    for (const auto & kv : doc.object) {
        if (!kv.second.isObject())
            throw std::runtime_error("material '" + kv.first + "' must be a JSON object");
        Surface s;
        s.specular = readBands(kv.second, "specular", kv.first);
        s.diffuse = readBands(kv.second, "diffuse", kv.first);
        ret[kv.first] = s;
    }
    return ret;
}

LoadedScene loadScene(const std::string & objpath, const std::string & materialFileName, bool verbose)
{
    LoadedScene scene;
    // rayverb.cpp:336-341: the built-in surface for unknown materials
    Surface def;
    const float spec[8] = {0.92f, 0.92f, 0.93f, 0.93f, 0.94f, 0.95f, 0.95f, 0.95f};
    const float diff[8] = {0.50f, 0.90f, 0.95f, 0.95f, 0.95f, 0.95f, 0.95f, 0.95f};
    for (int i = 0; i < 8; ++i) { def.specular.s[i] = spec[i]; def.diffuse.s[i] = diff[i]; }
    scene.surfaces.push_back(def);
    std::map<std::string, cl_ulong> materialIndices;
    for (const auto & kv : loadMaterials(materialFileName)) {       // std::map order, rayverb.cpp:348-354
        scene.surfaces.push_back(kv.second);
        scene.materialNames.push_back(kv.first);
        materialIndices[kv.first] = scene.surfaces.size() - 1;
    }

    std::ifstream in(objpath);
    if (!in)
        throw std::runtime_error("Failed to load object file.");    // rayverb.cpp:333-334
    std::string line;
    cl_ulong current = 0;
    std::string currentName;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#')
            continue;
        if (tag == "v") {
            cl_float3 v = {{0, 0, 0, 0}};
            ls >> v.s[0] >> v.s[1] >> v.s[2];
            scene.vertices.push_back(v);
        } else if (tag == "usemtl") {
            ls >> currentName;
            auto it = materialIndices.find(currentName);
            current = it == materialIndices.end() ? 0 : it->second;
            if (verbose)
                std::cerr << "Found mesh with material name: " << currentName << " -> surface " << current << std::endl;
        } else if (tag == "f") {
            std::vector<cl_ulong> poly;
            std::string item;
            while (ls >> item) {
                const long idx = std::strtol(item.c_str(), nullptr, 10);     // "v", "v/vt", "v//vn", "v/vt/vn"
                if (idx == 0)
                    throw std::runtime_error("Failed to load object file.");
                const long resolved = idx > 0 ? idx - 1 : (long) scene.vertices.size() + idx;
                if (resolved < 0 || (size_t) resolved >= scene.vertices.size())
                    throw std::runtime_error("Failed to load object file.");
                poly.push_back((cl_ulong) resolved);
            }
            triangulate(scene.vertices, poly, current, scene.triangles);
        }
    }
    if (verbose)
        std::cerr << "Loaded 3D model with " << scene.triangles.size() << " triangles" << std::endl;   // rayverb.cpp:437-444
    return scene;
}
