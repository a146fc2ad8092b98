#pragma once
#include <array>
#include <string>
#include <vector>

#include "hostmath.h"

namespace rth {

// What the reference's TriangleMesh holds before the object-to-world transform is applied
// (reference src/core/Shape/TriangleShape.cpp:18-55): per-corner vertices, normals, optional UVs.
struct ImportedMesh {
    std::vector<V3> position;
    std::vector<V3> normal;      // always filled (file normals or generated smooth normals)
    std::vector<float> uv;       // 2 per vertex, empty when the file has none
    std::vector<int> indices;    // 3 per triangle
};

// Throws std::runtime_error on unreadable input.
void import_obj(const std::string &path, ImportedMesh &out);

}  // namespace rth
