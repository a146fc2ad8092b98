// SAH KD-tree build on the host, producing the flat 8-byte-node arrays the device traverses.
// Same algorithm, constants, float arithmetic and tie handling as the reference's builder
// (reference src/core/Accelerators/KDTree.cpp:86-280; node encoding :13-21,46-69), so the node array
// equals what the reference's KdTree holds for the same primitive bounds (std::sort from the same
// C++ library decides the order of tied edges, KDTree.cpp:187-198).
//
// The reference builds on one thread. Here the two children of a large node near the root are built by two threads:
// the below child continues in the caller's arrays, the above child goes to a fragment of its own (own scratch, node and
// leaf-index offsets relative to the fragment), and the fragment is appended afterwards with its offsets shifted. Every
// subtree is still built by the sequential algorithm from the same primitive list in the same order, and nodes are laid
// out below-subtree first as the reference's m_nextFreeNode counter does, so the arrays do not depend on the thread count.
#include "kd_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <thread>

namespace rth {

namespace {

struct Edge {
    float t;
    int prim;
    int kind;   // 0 = start of the primitive's extent, 1 = end (sorted start-before-end on ties)
};

// Nodes and leaf indices of one subtree, offsets relative to the fragment.
struct Fragment {
    std::vector<uint32_t> nodes;
    std::vector<int32_t> leafIdx;
};

// Work arrays of one thread, sized for the subtree it builds.
// Left uninitialised: `above` is sized for the worst case and mostly never touched.
struct Scratch {
    std::unique_ptr<Edge[]> edges[3];
    std::unique_ptr<int[]> below;   // one primitive list, reused down the below-chain
    std::unique_ptr<int[]> above;   // (depth + 1) * n, a fresh window per level
    Scratch(size_t n, int depth) {
        for (auto &e : edges) e.reset(new Edge[2 * n + 2]);
        below.reset(new int[n + 1]);
        above.reset(new int[(size_t)(depth + 1) * n + 1]);
    }
};

constexpr int kSpawnMinPrims = 1 << 15;   // smaller nodes are not worth a thread

struct Builder {
    const std::vector<Box3> &primBounds;
    int isectCost = 80, travCost = 1, maxPrims = 1;
    float emptyBonus = 0.5f;

    explicit Builder(const std::vector<Box3> &pb) : primBounds(pb) {}

    static int newNode(Fragment &out) {
        out.nodes.push_back(0u);
        out.nodes.push_back(0u);
        return (int)(out.nodes.size() / 2) - 1;
    }

    static void makeLeaf(Fragment &out, int node, const int *prims, int n) {
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

    // Appends `f` (a subtree whose root is its node 0) to `out`; returns the index its root gets.
    static int append(Fragment &out, const Fragment &f) {
        const uint32_t nodeBase = (uint32_t)(out.nodes.size() / 2), leafBase = (uint32_t)out.leafIdx.size();
        out.nodes.reserve(out.nodes.size() + f.nodes.size());
        for (size_t i = 0; i < f.nodes.size(); i += 2) {
            uint32_t w0 = f.nodes[i], w1 = f.nodes[i + 1];
            if ((w1 & 3u) != 3u) w1 = (w1 & 3u) | (((w1 >> 2) + nodeBase) << 2);   // interior: above child
            else if ((w1 >> 2) > 1u) w0 += leafBase;                               // leaf with an index range
            out.nodes.push_back(w0);
            out.nodes.push_back(w1);
        }
        out.leafIdx.insert(out.leafIdx.end(), f.leafIdx.begin(), f.leafIdx.end());
        return (int)nodeBase;
    }

    // `spawn`: how many more levels may hand their above child to another thread.
    void build(Fragment &out, Scratch &sc, int node, const Box3 &nb, const int *prims, int n, int depth, int *belowRoom, int *aboveRoom,
               int badRefines, int spawn) {
        if (n <= maxPrims || depth == 0) {
            makeLeaf(out, node, prims, n);
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
            Edge *e = sc.edges[axis].get();
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
            makeLeaf(out, node, prims, n);
            return;
        }

        const Edge *be = sc.edges[bestAxis].get();
        int nb0 = 0, nb1 = 0;
        for (int i = 0; i < bestOffset; ++i)
            if (be[i].kind == 0) belowRoom[nb0++] = be[i].prim;
        for (int i = bestOffset + 1; i < 2 * n; ++i)
            if (be[i].kind == 1) aboveRoom[nb1++] = be[i].prim;

        const float split = be[bestOffset].t;
        Box3 b0 = nb, b1 = nb;
        b0.pMax[bestAxis] = split;
        b1.pMin[bestAxis] = split;

        // The above child's primitive list stays untouched in this level's window of `above` while the below subtree is
        // built (it works in the windows after it), so another thread may read it meanwhile.
        std::future<Fragment> other;
        if (spawn > 0 && nb1 >= kSpawnMinPrims && nb0 >= kSpawnMinPrims) {
            const int *abovePrims = aboveRoom;
            other = std::async(std::launch::async, [this, b1, abovePrims, nb1, depth, badRefines, spawn]() {
                Fragment f;
                Scratch own((size_t)nb1, depth - 1);
                newNode(f);
                build(f, own, 0, b1, abovePrims, nb1, depth - 1, own.below.get(), own.above.get(), badRefines, spawn - 1);
                return f;
            });
        }
        const int belowChild = newNode(out);   // == node + 1
        (void)belowChild;
        build(out, sc, node + 1, b0, belowRoom, nb0, depth - 1, belowRoom, aboveRoom + n, badRefines, spawn - 1);
        int aboveChild;
        if (other.valid()) {
            aboveChild = append(out, other.get());
        } else {
            aboveChild = newNode(out);
            build(out, sc, aboveChild, b1, aboveRoom, nb1, depth - 1, belowRoom, aboveRoom + n, badRefines, spawn - 1);
        }
        uint32_t w0;
        std::memcpy(&w0, &split, 4);
        out.nodes[2 * node] = w0;
        out.nodes[2 * node + 1] = (uint32_t)bestAxis | ((uint32_t)aboveChild << 2);
    }
};

}  // namespace

void kd_build(const std::vector<Box3> &primBounds, KdTreeFlat &out, int maxDepth, int threads) {
    out.nodes.clear();
    out.leafIdx.clear();
    const size_t n = primBounds.size();
    if (maxDepth <= 0) maxDepth = (int)std::round(8 + 1.3f * std::log2(float(int64_t(n))));
    out.maxDepthParam = maxDepth;

    Box3 bounds;   // empty: min = +max float, max = lowest float (Utils/Math.h:156-162)
    for (const Box3 &b : primBounds) bounds = box_union(bounds, b);
    out.bounds = bounds;

    if (threads <= 0) threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    // Levels whose above child goes to another thread. Subtrees of one level differ in work, so there are four times as
    // many of them as threads were asked for and the operating system balances them.
    int spawnLevels = 0;
    if (threads > 1) {
        while ((1 << spawnLevels) < threads) ++spawnLevels;
        spawnLevels += 2;
    }

    Builder bld(primBounds);
    Fragment whole;
    Scratch sc(n, maxDepth);
    std::vector<int> all(n);
    for (size_t i = 0; i < n; ++i) all[i] = (int)i;
    Builder::newNode(whole);
    bld.build(whole, sc, 0, bounds, all.data(), (int)n, maxDepth, sc.below.get(), sc.above.get(), 0, spawnLevels);
    out.nodes = std::move(whole.nodes);
    out.leafIdx = std::move(whole.leafIdx);
}

}  // namespace rth
