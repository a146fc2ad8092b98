#pragma once
#include <limits>
#include <vector>

#include "hostmath.h"

namespace rth {

// Axis-aligned box; the default is the "empty" box of the reference's BBox3() (Utils/Math.h:156-162).
struct Box3 {
    V3 pMin{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    V3 pMax{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
};
inline float fmin_glm(float x, float y) { return (y < x) ? y : x; }   // glm::min
inline float fmax_glm(float x, float y) { return (x < y) ? y : x; }   // glm::max
inline Box3 box_union(const Box3 &a, const Box3 &b) {
    Box3 r;
    r.pMin = {fmin_glm(a.pMin.x, b.pMin.x), fmin_glm(a.pMin.y, b.pMin.y), fmin_glm(a.pMin.z, b.pMin.z)};
    r.pMax = {fmax_glm(a.pMax.x, b.pMax.x), fmax_glm(a.pMax.y, b.pMax.y), fmax_glm(a.pMax.z, b.pMax.z)};
    return r;
}
inline Box3 box_union(const Box3 &a, V3 p) {
    Box3 r;
    r.pMin = {fmin_glm(a.pMin.x, p.x), fmin_glm(a.pMin.y, p.y), fmin_glm(a.pMin.z, p.z)};
    r.pMax = {fmax_glm(a.pMax.x, p.x), fmax_glm(a.pMax.y, p.y), fmax_glm(a.pMax.z, p.z)};
    return r;
}
inline Box3 box_of(V3 a, V3 b) {
    Box3 r;
    r.pMin = {fmin_glm(a.x, b.x), fmin_glm(a.y, b.y), fmin_glm(a.z, b.z)};
    r.pMax = {fmax_glm(a.x, b.x), fmax_glm(a.y, b.y), fmax_glm(a.z, b.z)};
    return r;
}

struct KdTreeFlat {
    std::vector<uint32_t> nodes;   // 2 words per node
    std::vector<int32_t> leafIdx;
    Box3 bounds;
    int maxDepthParam = 0;
};

// maxDepth <= 0 selects round(8 + 1.3 log2 N) like the reference (Accelerators/KDTree.cpp:95-98). threads <= 0: as many as
// the machine has (at most 32); the arrays are the same for every thread count.
void kd_build(const std::vector<Box3> &primBounds, KdTreeFlat &out, int maxDepth = -1, int threads = 0);

}  // namespace rth
