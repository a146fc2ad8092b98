// SAH KD-tree build on the host, producing the flat 8-byte-node arrays the device traverses.
// Same algorithm, constants, float arithmetic and tie handling as the reference's builder
// (reference src/core/Accelerators/KDTree.cpp:86-280; node encoding :13-21,46-69), so the node array
// equals what the reference's KdTree holds for the same primitive bounds (std::sort from the same
// C++ library decides the order of tied edges, KDTree.cpp:187-198).
#include "kd_build.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rth {

namespace {

struct Edge {
    float t;
    int prim;
    int kind;   // 0 = start of the primitive's extent, 1 = end (sorted start-before-end on ties)
};

struct Builder {
    const std::vector<Box3> &primBounds;
    KdTreeFlat &out;
    int isectCost = 80, travCost = 1, maxPrims = 1;
    float emptyBonus = 0.5f;
    std::vector<Edge> edges[3];
    std::vector<int> below;   // one primitive list, reused down the below-chain
    std::vector<int> above;   // (maxDepth + 1) * N, a fresh window per level

    Builder(const std::vector<Box3> &pb, KdTreeFlat &o) : primBounds(pb), out(o) {}

    int newNode() {
        out.nodes.push_back(0u);
        out.nodes.push_back(0u);
        return (int)(out.nodes.size() / 2) - 1;
    }

    void makeLeaf(int node, const int *prims, int n) {
        uint32_t w0 = 0;
        if (n == 1) {
            w0 = (uint32_t)prims[0];
        } else if (n > 1) {
            w0 = (uint32_t)out.leafIdx.size();
            for (int i = 0; i < n; ++i) out.leafIdx.push_back(prims[i]);
        }
        out.nodes[2 * node] = w0;
        out.nodes[2 * node + 1] = 3u | ((uint32_t)n << 2);
    }

    void build(int node, const Box3 &nb, const int *prims, int n, int depth, int *belowRoom, int *aboveRoom, int badRefines) {
        if (n <= maxPrims || depth == 0) {
            makeLeaf(node, prims, n);
            return;
        }
        int bestAxis = -1, bestOffset = -1;
        float bestCost = std::numeric_limits<float>::infinity();
        const float oldCost = isectCost * float(n);
        const V3 d = nb.pMax - nb.pMin;
        const float totalSA = 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
        const float invTotalSA = 1 / totalSA;

        int axis = (d.x > d.y && d.x > d.z) ? 0 : (d.y > d.z ? 1 : 2);
        for (int retries = 0;; ++retries) {
            Edge *e = edges[axis].data();
            for (int i = 0; i < n; ++i) {
                const int pi = prims[i];
                const Box3 &b = primBounds[pi];
                e[2 * i] = Edge{b.pMin[axis], pi, 0};
                e[2 * i + 1] = Edge{b.pMax[axis], pi, 1};
            }
            std::sort(e, e + 2 * n, [](const Edge &a, const Edge &b) -> bool {
                if (a.t == b.t) return a.kind < b.kind;
                return a.t < b.t;
            });

            int nBelow = 0, nAbove = n;
            for (int i = 0; i < 2 * n; ++i) {
                if (e[i].kind == 1) --nAbove;
                const float t = e[i].t;
                if (t > nb.pMin[axis] && t < nb.pMax[axis]) {
                    const int a0 = (axis + 1) % 3, a1 = (axis + 2) % 3;
                    const float belowSA = 2 * (d[a0] * d[a1] + (t - nb.pMin[axis]) * (d[a0] + d[a1]));
                    const float aboveSA = 2 * (d[a0] * d[a1] + (nb.pMax[axis] - t) * (d[a0] + d[a1]));
                    const float pBelow = belowSA * invTotalSA;
                    const float pAbove = aboveSA * invTotalSA;
                    const float eb = (nAbove == 0 || nBelow == 0) ? emptyBonus : 0;
                    const float cost = travCost + isectCost * (1 - eb) * (pBelow * nBelow + pAbove * nAbove);
                    if (cost < bestCost) {
                        bestCost = cost;
                        bestAxis = axis;
                        bestOffset = i;
                    }
                }
                if (e[i].kind == 0) ++nBelow;
            }
            if (bestAxis == -1 && retries < 2) {
                axis = (axis + 1) % 3;
                continue;
            }
            break;
        }

        if (bestCost > oldCost) ++badRefines;
        if ((bestCost > 4 * oldCost && n < 16) || bestAxis == -1 || badRefines == 3) {
            makeLeaf(node, prims, n);
            return;
        }

        const Edge *be = edges[bestAxis].data();
        int nb0 = 0, nb1 = 0;
        for (int i = 0; i < bestOffset; ++i)
            if (be[i].kind == 0) belowRoom[nb0++] = be[i].prim;
        for (int i = bestOffset + 1; i < 2 * n; ++i)
            if (be[i].kind == 1) aboveRoom[nb1++] = be[i].prim;

        const float split = be[bestOffset].t;
        Box3 b0 = nb, b1 = nb;
        b0.pMax[bestAxis] = split;
        b1.pMin[bestAxis] = split;

        const int belowChild = newNode();   // == node + 1
        (void)belowChild;
        build(node + 1, b0, belowRoom, nb0, depth - 1, belowRoom, aboveRoom + n, badRefines);
        const int aboveChild = newNode();
        uint32_t w0;
        std::memcpy(&w0, &split, 4);
        out.nodes[2 * node] = w0;
        out.nodes[2 * node + 1] = (uint32_t)bestAxis | ((uint32_t)aboveChild << 2);
        build(aboveChild, b1, aboveRoom, nb1, depth - 1, belowRoom, aboveRoom + n, badRefines);
    }
};

}  // namespace

void kd_build(const std::vector<Box3> &primBounds, KdTreeFlat &out, int maxDepth) {
    out.nodes.clear();
    out.leafIdx.clear();
    const size_t n = primBounds.size();
    if (maxDepth <= 0) maxDepth = (int)std::round(8 + 1.3f * std::log2(float(int64_t(n))));
    out.maxDepthParam = maxDepth;

    Box3 bounds;   // empty: min = +max float, max = lowest float (Utils/Math.h:156-162)
    for (const Box3 &b : primBounds) bounds = box_union(bounds, b);
    out.bounds = bounds;

    Builder bld(primBounds, out);
    for (int a = 0; a < 3; ++a) bld.edges[a].resize(2 * n + 2);
    bld.below.resize(n + 1);
    bld.above.resize((size_t)(maxDepth + 1) * n + 1);
    std::vector<int> all(n);
    for (size_t i = 0; i < n; ++i) all[i] = (int)i;

    bld.newNode();
    bld.build(0, bounds, all.data(), (int)n, maxDepth, bld.below.data(), bld.above.data(), 0);
}

}  // namespace rth
