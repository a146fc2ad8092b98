// Issue-slot model of ray-to-lane schedules for the KD-tree tracer on a scene whose tables sit in LDS. Every ray of a batch
// is traced on the CPU with the device headers themselves (dtraverse.h against the stand-in tests/emu/hip/hip_runtime.h, so
// node visits, tests and arithmetic are the kernels'), its sequence of leaf visits is recorded (node steps before the leaf,
// primitives in it, which of the tests got past the sign rejection), and the wave-level instruction count of a schedule is the
// sum over its loop iterations of the instructions an iteration issues when at least one lane needs it. Constants come from
// the ISA of k_extend<true, false, false>. A tool for choosing what to build; nothing in the library depends on it.
//
//   g++ -O2 -std=c++17 -ffp-contract=off -Itests/emu -o /tmp/sched_sim tools/sched_sim.cpp && /tmp/sched_sim scene.bin rays.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../raytracing_amd/csrc/device/dtraverse.h"

namespace {

struct Visit {
    int desc;         // interior nodes stepped through before this leaf
    int n;            // primitives in the leaf
    uint32_t cand;    // bit i: test i got past the edge-sign rejection (runs the long tail of the test)
};
struct RayTrace {
    std::vector<Visit> v;
    bool clipped;     // missed the root box: no traversal
};

template <typename T> std::vector<T> rd(FILE *f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

// kd_trace<CLOSEST> with the events written down (tables in HBM flavour of the arithmetic: PERM = false gives the same values)
RayTrace trace_events(const DevScene &sc, const SceneTables &tb, f3 o, f3 d) {
    RayTrace out;
    RayCtx r;
    ray_setup(r, o, d);
    float tMin, tMax, rayTMax = RT_INF;
    out.clipped = !bounds_clip(sc, r, rayTMax, tMin, tMax);
    if (out.clipped) return out;
    uint32_t stN[64];
    float stT[64];
    int sp = 0;
    uint32_t node = 0;
    for (;;) {
        if (rayTMax < tMin) break;
        Visit vis = {0, 0, 0};
        uint2 nd = tb.nodes[node];
        while ((nd.y & 3u) != 3u) {
            ++vis.desc;
            const int axis = (int)(nd.y & 3u);
            const float split = __uint_as_float(nd.x);
            const float oa = comp3(r.o, axis), da = comp3(r.d, axis), ia = comp3(r.invDir, axis);
            const float tPlane = (split - oa) * ia;
            const bool belowFirst = (oa < split) || (oa == split && da <= 0);
            const uint32_t above = nd.y >> 2;
            const uint32_t first = belowFirst ? node + 1 : above;
            const uint32_t second = belowFirst ? above : node + 1;
            if (tPlane > tMax || tPlane <= 0) node = first;
            else if (tPlane < tMin) node = second;
            else {
                stN[sp] = second; stT[sp] = tMax; ++sp;
                node = first;
                tMax = tPlane;
            }
            nd = tb.nodes[node];
        }
        const int n = (int)(nd.y >> 2);
        vis.n = n;
        for (int i = 0; i < n; ++i) {
            const int prim = n == 1 ? (int)nd.x : tb.leafIdx[nd.x + i];
            const float4 q0 = tb.triHot[3 * prim], q1 = tb.triHot[3 * prim + 1], q2 = tb.triHot[3 * prim + 2];
            // candidate = passes the sign test (recomputed here as tri_test does)
            {
                f3 p0t = perm3(mk3(q0.x, q0.y, q0.z), r.kz) - r.op, p1t = perm3(mk3(q1.x, q1.y, q1.z), r.kz) - r.op, p2t = perm3(mk3(q2.x, q2.y, q2.z), r.kz) - r.op;
                p0t.x += r.Sx * p0t.z; p0t.y += r.Sy * p0t.z;
                p1t.x += r.Sx * p1t.z; p1t.y += r.Sy * p1t.z;
                p2t.x += r.Sx * p2t.z; p2t.y += r.Sy * p2t.z;
                const float e0 = p1t.x * p2t.y - p1t.y * p2t.x, e1 = p2t.x * p0t.y - p2t.y * p0t.x, e2 = p0t.x * p1t.y - p0t.y * p1t.x;
                if (!((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0))) vis.cand |= 1u << i;
            }
            float t, b0, b1, b2;
            if (tri_test<false>(r, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z), rayTMax, t, b0, b1, b2) && q0.w != RT_PRIM_DEGENERATE) rayTMax = t;
        }
        out.v.push_back(vis);
        if (sp == 0) break;
        --sp;
        tMin = tMax;
        node = stN[sp];
        tMax = stT[sp];
    }
    return out;
}

// instruction counts (vector + LDS instructions a wave issues per loop iteration), from the ISA
double cS = 150;    // ray fetch, set-up (7 divisions), root clip, result store: once per ray, all lanes together
double cN = 26;     // one interior node step
double cT = 46;     // one triangle test that the edge signs reject
double cC = 58;     // the rest of a test that got past them
double cL = 22;     // leaf header, todo pop, loop bookkeeping per leaf visit
double cPop = 24;   // taking prepared rays from the wave's pool (streaming forms)

int popc(uint32_t x) { return __builtin_popcount(x); }

struct Totals {
    double issue = 0;
    double rays = 0;
};

// current kernel: 64 rays per wave, lanes meet at every leaf
double lockstep_wave(const std::vector<RayTrace> &rays, size_t b, size_t e) {
    double c = cS;
    size_t maxV = 0;
    for (size_t i = b; i < e; ++i) maxV = std::max(maxV, rays[i].v.size());
    for (size_t k = 0; k < maxV; ++k) {
        int md = 0, mn = 0;
        uint32_t candIter = 0;
        for (size_t i = b; i < e; ++i)
            if (k < rays[i].v.size()) {
                md = std::max(md, rays[i].v[k].desc);
                mn = std::max(mn, rays[i].v[k].n);
                candIter |= rays[i].v[k].cand;
            }
        c += md * cN + mn * cT + popc(candIter) * cC + cL;
    }
    return c;
}

// lanes meet at every leaf as now, but a wave works through a stream of rays and lanes whose ray has ended take a prepared ray
// at the next round boundary (prepared 64 at a time at full width when at least `refillMin` lanes are idle)
double lockstep_refill(const std::vector<RayTrace> &rays, int refillMin, double *roundsOut = nullptr) {
    double c = 0, rounds = 0;
    size_t next = 0;
    const size_t n = rays.size();
    int cur[64];
    size_t pos[64];
    for (int l = 0; l < 64; ++l) cur[l] = -1;
    size_t pool = 0;   // prepared rays waiting
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) idle += cur[l] < 0;
        if (idle >= refillMin || idle == 64) {
            if (pool == 0 && next < n) {
                pool = std::min<size_t>(64, n - next);
                c += cS;
            }
            if (pool > 0) {
                c += cPop;
                for (int l = 0; l < 64 && pool > 0; ++l)
                    if (cur[l] < 0) {
                        // rays that miss the root box never enter a lane
                        while (pool > 0 && rays[next].clipped) { ++next; --pool; }
                        if (pool == 0) break;
                        cur[l] = (int)next; pos[l] = 0; ++next; --pool;
                    }
            }
        }
        int md = 0, mn = 0, act = 0;
        uint32_t candIter = 0;
        for (int l = 0; l < 64; ++l)
            if (cur[l] >= 0) {
                const Visit &v = rays[cur[l]].v[pos[l]];
                md = std::max(md, v.desc); mn = std::max(mn, v.n); candIter |= v.cand;
                ++act;
            }
        if (act == 0) {
            if (next >= n && pool == 0) break;
            continue;
        }
        c += md * cN + mn * cT + popc(candIter) * cC + cL;
        ++rounds;
        for (int l = 0; l < 64; ++l)
            if (cur[l] >= 0 && ++pos[l] >= rays[cur[l]].v.size()) cur[l] = -1;
    }
    if (roundsOut) *roundsOut = rounds;
    return c;
}

// one loop: every iteration runs a node step for the lanes between leaves and a test for the lanes inside one
double ifif_stream(const std::vector<RayTrace> &rays, int refillMin, int dMaj, int tMaj) {
    // dMaj / tMaj > 0: the phase form (repeat node steps while >= dMaj lanes want one, then tests while >= tMaj want one)
    double c = 0;
    size_t next = 0, pool = 0;
    const size_t n = rays.size();
    struct Lane { int ray = -1; size_t vis = 0; int descLeft = 0, testI = 0; } L[64];
    auto enter = [&](Lane &l) {   // start visit l.vis, or end the ray
        if (l.vis >= rays[l.ray].v.size()) { l.ray = -1; return; }
        l.descLeft = rays[l.ray].v[l.vis].desc;
        l.testI = 0;
    };
    auto wantsNode = [&](const Lane &l) { return l.ray >= 0 && l.descLeft > 0; };
    // a lane that reaches its leaf reads the leaf node in a node step too (the step that finds the leaf flag)
    auto stepNode = [&](Lane &l) { --l.descLeft; };
    auto wantsTest = [&](const Lane &l) { return l.ray >= 0 && l.descLeft == 0; };
    for (;;) {
        int idle = 0;
        for (auto &l : L) idle += l.ray < 0;
        if (idle >= refillMin || idle == 64) {
            if (pool == 0 && next < n) { pool = std::min<size_t>(64, n - next); c += cS; }
            if (pool > 0) {
                c += cPop;
                for (auto &l : L)
                    if (l.ray < 0) {
                        while (pool > 0 && rays[next].clipped) { ++next; --pool; }
                        if (pool == 0) break;
                        l.ray = (int)next; l.vis = 0; ++next; --pool;
                        enter(l);
                    }
            }
        }
        int act = 0;
        for (auto &l : L) act += l.ray >= 0;
        if (act == 0) {
            if (next >= n && pool == 0) break;
            continue;
        }
        // node phase
        for (;;) {
            int w = 0;
            for (auto &l : L) w += wantsNode(l);
            if (w == 0) break;
            c += cN + 4;
            for (auto &l : L) if (wantsNode(l)) stepNode(l);
            int w2 = 0;
            for (auto &l : L) w2 += wantsNode(l);
            if (w2 < dMaj) break;
        }
        // test phase
        for (;;) {
            int w = 0;
            bool cand = false;
            for (auto &l : L) w += wantsTest(l);
            if (w == 0) break;
            for (auto &l : L)
                if (wantsTest(l)) {
                    const Visit &v = rays[l.ray].v[l.vis];
                    if (l.testI < v.n) {
                        cand |= (v.cand >> l.testI) & 1u;
                        ++l.testI;
                    }
                    if (l.testI >= v.n) { ++l.vis; enter(l); }
                }
            c += cT + 8 + (cand ? cC : 0);
            int w2 = 0;
            for (auto &l : L) w2 += wantsTest(l);
            if (w2 < tMaj) break;
        }
    }
    return c;
}

// lanes meet at leaves; the tests of a round are pooled: all (ray, primitive) pairs of the round are spread over the 64 lanes
double pooled(const std::vector<RayTrace> &rays, int refillMin, double cPair, double cRound, double cResolve) {
    double c = 0;
    size_t next = 0, pool = 0;
    const size_t n = rays.size();
    int cur[64];
    size_t pos[64];
    for (int l = 0; l < 64; ++l) cur[l] = -1;
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) idle += cur[l] < 0;
        if (idle >= refillMin || idle == 64) {
            if (pool == 0 && next < n) { pool = std::min<size_t>(64, n - next); c += cS; }
            if (pool > 0) {
                c += cPop;
                for (int l = 0; l < 64 && pool > 0; ++l)
                    if (cur[l] < 0) {
                        while (pool > 0 && rays[next].clipped) { ++next; --pool; }
                        if (pool == 0) break;
                        cur[l] = (int)next; pos[l] = 0; ++next; --pool;
                    }
            }
        }
        int md = 0, act = 0, pairs = 0, maxCand = 0;
        for (int l = 0; l < 64; ++l)
            if (cur[l] >= 0) {
                const Visit &v = rays[cur[l]].v[pos[l]];
                md = std::max(md, v.desc); pairs += v.n; maxCand = std::max(maxCand, popc(v.cand));
                ++act;
            }
        if (act == 0) {
            if (next >= n && pool == 0) break;
            continue;
        }
        c += md * cN + cL + cRound + ((pairs + 63) / 64) * cPair + maxCand * cResolve;
        for (int l = 0; l < 64; ++l)
            if (cur[l] >= 0 && ++pos[l] >= rays[cur[l]].v.size()) cur[l] = -1;
    }
    return c;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: sched_sim scene.bin rays.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const auto hdr = rd<uint32_t>(f, 7);
    rd<int32_t>(f, 10);
    rd<uint64_t>(f, 1);
    const auto bounds = rd<float>(f, 6);
    rd<float>(f, 32);
    const uint32_t nNodes = hdr[1], nIdx = hdr[2], nPrims = hdr[3];
    const auto nodes = rd<uint32_t>(f, (size_t)nNodes * 2);
    const auto idx = rd<int32_t>(f, nIdx);
    const auto triP = rd<float>(f, (size_t)nPrims * 9);
    fclose(f);
    std::vector<float4> hot((size_t)nPrims * 3);
    for (uint32_t p = 0; p < nPrims; ++p)
        for (int k = 0; k < 3; ++k) hot[3 * p + k] = make_float4(triP[9 * p + 3 * k], triP[9 * p + 3 * k + 1], triP[9 * p + 3 * k + 2], 0.f);
    DevScene sc = {};
    for (int i = 0; i < 3; ++i) { sc.bmin[i] = bounds[i]; sc.bmax[i] = bounds[3 + i]; }
    sc.nNodes = nNodes; sc.nLeafIdx = nIdx; sc.nPrims = nPrims; sc.triCopies = 1;
    SceneTables tb = {reinterpret_cast<const uint2 *>(nodes.data()), idx.data(), hot.data()};

    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    const uint64_t n = rd<uint64_t>(f, 1)[0];
    const auto o = rd<float>(f, n * 3), d = rd<float>(f, n * 3);
    fclose(f);
    std::vector<RayTrace> rays(n);
    double nodesT = 0, testsT = 0, visitsT = 0, candT = 0, clipped = 0;
    std::vector<double> histV(40, 0);
    for (uint64_t i = 0; i < n; ++i) {
        rays[i] = trace_events(sc, tb, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        clipped += rays[i].clipped;
        visitsT += rays[i].v.size();
        histV[std::min<size_t>(39, rays[i].v.size())] += 1;
        for (auto &v : rays[i].v) { nodesT += v.desc; testsT += v.n; candT += popc(v.cand); }
    }
    printf("%llu rays: %.2f node steps, %.2f leaf visits, %.2f tests, %.2f past the sign test per ray; %.3f miss the root box\n",
           (unsigned long long)n, nodesT / n, visitsT / n, testsT / n, candT / n, clipped / n);
    printf("leaf visits per ray:");
    for (int k = 0; k < 16; ++k) printf(" %d:%.3f", k, histV[k] / n);
    printf("\n");
    const double useful = (n * cS + nodesT * cN + testsT * cT + candT * cC + visitsT * cL) / 64.0;
    printf("instructions if every lane were busy: %.1f per ray\n", useful / n);
    // lockstep, waves of 64 consecutive rays; per-round occupancy
    double lock = 0;
    std::vector<double> roundAct(24, 0), roundCnt(24, 0);
    for (uint64_t b = 0; b < n; b += 64) {
        const size_t e = std::min<uint64_t>(n, b + 64);
        lock += lockstep_wave(rays, b, e);
        for (size_t k = 0; k < 24; ++k) {
            int a = 0;
            for (size_t i = b; i < e; ++i) a += k < rays[i].v.size();
            if (a) { roundAct[k] += a; roundCnt[k] += 1; }
        }
    }
    printf("lockstep (now): %.1f instructions per ray, lane use %.1f of 64\n", lock / n, 64 * useful / lock);
    printf("  waves still running in round k and their mean live lanes:");
    for (int k = 0; k < 12; ++k) if (roundCnt[k] > 0) printf(" %d:%.2f/%.1f", k, roundCnt[k] / ((n + 63) / 64), roundAct[k] / roundCnt[k]);
    printf("\n");
    for (int rm : {1, 8, 16, 32}) {
        double rounds = 0;
        const double c = lockstep_refill(rays, rm, &rounds);
        printf("lockstep + refill at leaf rounds (>= %2d idle): %.1f per ray, lane use %.1f\n", rm, c / n, 64 * useful / c);
    }
    for (auto pr : {std::pair<int, int>{1, 1}, {8, 24}, {16, 16}, {24, 32}, {32, 32}, {64, 64}}) {
        const double c = ifif_stream(rays, 4, pr.first, pr.second);
        printf("phases (node while >= %2d, test while >= %2d), refill: %.1f per ray, lane use %.1f\n", pr.first, pr.second, c / n, 64 * useful / c);
    }
    for (int rm : {1, 16}) {
        const double c = pooled(rays, rm, cT + 34, 60, 14);
        printf("pooled tests + refill (>= %2d idle): %.1f per ray, lane use %.1f\n", rm, c / n, 64 * useful / c);
    }
    return 0;
}
