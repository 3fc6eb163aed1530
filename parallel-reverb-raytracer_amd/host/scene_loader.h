// scene_loader.h — Wavefront OBJ + JSON materials -> triangles / vertices / surfaces, the job of
// the reference's Raytracer::SceneData (rayverb/rayverb.cpp:296-507, which delegates the parsing
// to Assimp).  Host-side, once per run.
#pragma once

#include "../../include/rayverb/clstructs.h"

#include <map>
#include <string>
#include <vector>

struct LoadedScene {
    std::vector<Triangle> triangles;
    std::vector<cl_float3> vertices;
    std::vector<Surface> surfaces;
    std::vector<std::string> materialNames;     // materialNames[i] belongs to surfaces[i + 1]
};

// name -> Surface, from a JSON object of {"specular": [8 numbers], "diffuse": [8 numbers]} entries
std::map<std::string, Surface> loadMaterials(const std::string & materialFileName);

// Throws std::runtime_error when a file cannot be read or parsed.
LoadedScene loadScene(const std::string & objpath, const std::string & materialFileName, bool verbose);
