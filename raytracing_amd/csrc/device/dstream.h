// Streaming KD-tree tracer: one wave works through a stream of rays and keeps its 64 lanes busy.
//
// kd_trace (dtraverse.h) gives one ray to one lane and the wave waits for its slowest lane: on rays that
// leave surfaces in random directions only about one lane in five does useful work per instruction. Here
// a lane that finishes its ray takes the next prepared ray from a per-wave pool in LDS, and the wave
// refills that pool 64 rays at a time with all lanes (ray set-up and the root clip need six divisions, so
// they are done at full width, never by a few idle lanes). Between refills the wave alternates between
// rounds of one interior-node step and rounds of one triangle test per lane; a phase is repeated only while
// most lanes still want it, so coherent rays run as two tight loops and incoherent rays do not wait for
// the lane with the deepest descent or the biggest leaf. Per ray the node visits, primitive tests and float operations are exactly
// those of kd_trace, hence of the reference's KdTree::hit pair (src/core/Accelerators/KDTree.cpp:284-490);
// only the order in which different rays make progress changes.
//
// The Job type supplies the rays and consumes the results:
//   bool fetch(uint32_t j, f3 &o, f3 &d, float &tMax)   ray j of the stream (false: entry j has no ray)
//   void finish(uint32_t j, const HitRec &h)            called exactly once per fetched ray by the lane that
//                                                       ended it; h.prim < 0 is a miss, h.t then is tMax
#pragma once
#include "dtraverse.h"

// Phase thresholds. A node step out of LDS costs little next to a primitive test, so a wave keeps descending with few lanes
// (8) and switches to the tests when most lanes wait for one; a node step out of HBM costs a memory round trip whatever the
// number of lanes, so there the wave leaves the descent as soon as fewer than 24 lanes want it. Measured, 8 / 24 against
// 24 / 16: cornellBox bounce rays 14.9 / 13.8 G rays/s; 1 M-triangle soup, camera rays 1.30 / 1.65 G rays/s, path tracing
// 429 / 474 M samples/s; 4 M-triangle mesh, incoherent rays 1.18 / 1.36 G rays/s.
#ifndef RT_D_MAJ_LDS
#define RT_D_MAJ_LDS 8       // tables in LDS: keep descending while at least this many lanes are between leaves
#endif
#ifndef RT_T_MAJ_LDS
#define RT_T_MAJ_LDS 24      // tables in LDS: keep testing while at least this many lanes have leaf primitives left
#endif
#ifndef RT_D_MAJ
#define RT_D_MAJ 24          // tables in HBM
#endif
#ifndef RT_T_MAJ
#define RT_T_MAJ 16          // tables in HBM
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 16     // idle lanes needed before the wave prepares the next 64 rays
#endif
#ifndef RT_POP_MIN
#define RT_POP_MIN 4         // idle lanes needed before prepared rays are handed out
#endif


// Tables in HBM: 1 = every round serves rays in all three states (node step, leaf index, primitive test) behind one wait;
// 0 = the two phases below.
#ifndef RT_STREAM_MERGED
#define RT_STREAM_MERGED 0
#endif
#ifndef RT_MERGED_T_MIN
#define RT_MERGED_T_MIN 1    // merged rounds: lanes that must wait for a primitive test before the wave runs one
#endif

#define RT_POOL_VEC_PER_RAY 4                                   // uint4 per prepared ray
#define RT_POOL_BYTES (RT_BLOCK * RT_POOL_VEC_PER_RAY * 16)     // per workgroup: 64 prepared rays per wave

// Prepared ray in the pool: {o, d.x} {d.yz, invDir.xy} {invDir.z, Sx, Sy, Sz} {tMin, tMax, rayTMax, job}
template <bool CLOSEST, bool STATS, int SL, bool PERM, bool SPH, class Job>
RT_DEV void trace_stream(const DevScene &sc, const SceneTables &tb, uint32_t *ldsStack, uint4 *ldsPool, uint32_t n, Job &job, TravCounters &cn) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint4 *pool = ldsPool + wave * (64 * RT_POOL_VEC_PER_RAY);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t nChunks = (n + 63u) >> 6;
    // Rays are taken in chunks of 64, chunk c by wave (c mod number of waves): the launch keeps every workgroup
    // resident, so all waves run from the first ray to the last and equal shares are balanced shares.
    const uint32_t chunkStride = gridDim.x * (RT_BLOCK / 64);
    uint32_t chunk = blockIdx.x * (RT_BLOCK / 64) + wave;
    uint32_t poolCount = 0;

    uint32_t privNode[RT_STACK_TOTAL - SL];
    float privTMax[RT_STACK_TOTAL - SL];

    // per-lane ray state
    RayCtx r;
    r.o = r.d = r.invDir = r.op = mk3(0.f, 0.f, 0.f);
    r.kz = 2;
    r.Sx = r.Sy = r.Sz = 0.f;
    float rayTMax = 0.f, tMin = 0.f, tMax = 0.f;
    uint32_t node = 0, jobId = 0, leafBase = 0, triBase = 0;
    int sp = 0, leafI = 0, leafN = 0;
    int prim = -1;   // merged rounds: the primitive to test next, -1 while its index is still to be read
    constexpr bool MERGED = RT_STREAM_MERGED && !PERM;
    HitRec hit = {-1, 0.f, 0.f, 0.f, 0.f};
    bool active = false;

    // next todo entry; false when the ray is finished (stack empty, or the closest hit so far lies before the entry)
    auto pop = [&]() -> bool {
        if (sp == 0) return false;
        --sp;
        tMin = tMax;   // the segment after the one just finished starts where that one ended
        // the LDS part is read unconditionally (clamped index) and the private part under its own branch: a single
        // load through a pointer that may point to either becomes a flat load, which waits on both memory paths
        const int e = sp < SL ? sp : SL - 1;
        node = ldsStack[(e * 2 + 0) * RT_BLOCK + tid];
        tMax = __uint_as_float(ldsStack[(e * 2 + 1) * RT_BLOCK + tid]);
        RT_PIN_VGPRS(node, tMax);   // keeps the two LDS reads as LDS reads (no merge with the branch below)
        if (sp >= SL) {
            const int q = sp - SL;
            node = privNode[q];
            tMax = privTMax[q];
        }
        if (CLOSEST && rayTMax < tMin) return false;
        return true;
    };

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (nIdle >= RT_POP_MIN) {
            if (poolCount == 0 && chunk < nChunks && nIdle >= RT_REFILL_MIN) {
                // all 64 lanes prepare one ray each, whatever they were doing
                const uint32_t j = chunk * 64u + lane;
                chunk += chunkStride;
                bool valid = false;
                RayCtx nr;
                float ntm = 0.f, t0 = 0.f, t1 = 0.f;
                if (j < n) {
                    f3 o, d;
                    if (job.fetch(j, o, d, ntm)) {
                        if (STATS) ++cn.rays;
                        ray_setup(nr, o, d);
                        valid = bounds_clip(sc, nr, ntm, t0, t1);
                        if (!valid) {
                            const HitRec miss = {-1, ntm, 0.f, 0.f, 0.f};
                            job.finish(j, miss);
                        }
                    }
                }
                const unsigned long long vm = __ballot(valid);
                if (valid) {
                    uint4 *e = pool + (uint32_t)__popcll(vm & below) * RT_POOL_VEC_PER_RAY;
                    e[0] = make_uint4(__float_as_uint(nr.o.x), __float_as_uint(nr.o.y), __float_as_uint(nr.o.z), __float_as_uint(nr.d.x));
                    e[1] = make_uint4(__float_as_uint(nr.d.y), __float_as_uint(nr.d.z), __float_as_uint(nr.invDir.x), __float_as_uint(nr.invDir.y));
                    e[2] = make_uint4(__float_as_uint(nr.invDir.z), __float_as_uint(nr.Sx), __float_as_uint(nr.Sy), __float_as_uint(nr.Sz));
                    e[3] = make_uint4(__float_as_uint(t0), __float_as_uint(t1), __float_as_uint(ntm), j);
                }
                poolCount = (uint32_t)__popcll(vm);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (poolCount > 0) {
                const uint32_t rank = (uint32_t)__popcll(idle & below);
                if (!active && rank < poolCount) {
                    const uint4 *e = pool + (poolCount - 1u - rank) * RT_POOL_VEC_PER_RAY;
                    const uint4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
                    r.o = mk3(__uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e0.z));
                    r.d = mk3(__uint_as_float(e0.w), __uint_as_float(e1.x), __uint_as_float(e1.y));
                    r.invDir = mk3(__uint_as_float(e1.z), __uint_as_float(e1.w), __uint_as_float(e2.x));
                    r.Sx = __uint_as_float(e2.y);
                    r.Sy = __uint_as_float(e2.z);
                    r.Sz = __uint_as_float(e2.w);
                    const f3 a = mk3(fabsf(r.d.x), fabsf(r.d.y), fabsf(r.d.z));
                    r.kz = (a.x > a.y) ? ((a.x > a.z) ? 0 : 2) : ((a.y > a.z) ? 1 : 2);
                    r.op = perm3(r.o, r.kz);
                    triBase = tri_table_base<PERM>(sc, r.kz);
                    tMin = __uint_as_float(e3.x);
                    tMax = __uint_as_float(e3.y);
                    rayTMax = __uint_as_float(e3.z);
                    jobId = e3.w;
                    node = 0;
                    sp = 0;
                    leafI = leafN = 0;
                    hit.prim = -1;
                    hit.t = rayTMax;
                    hit.b0 = hit.b1 = hit.b2 = 0.f;
                    active = true;
                    if (CLOSEST && rayTMax < tMin) {
                        job.finish(jobId, hit);
                        active = false;
                    }
                }
                poolCount -= nIdle < poolCount ? nIdle : poolCount;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if (nIdle == 64u && chunk >= nChunks) {
                break;
            }
        }

        if constexpr (MERGED) {
            // One round for all lanes: every ray issues the one load it needs next - a node, the first index of a leaf, or a
            // primitive record together with the index of the leaf's next primitive - and the wave waits once for all of them.
            // Lanes whose ray is in another state read entry 0 of the table (one cached line).
            const bool wantNode = active && leafI >= leafN;
            const bool wantIdx = active && leafI < leafN && prim < 0;
            bool wantTest = active && leafI < leafN && prim >= 0;
            // The test is the expensive section and runs at the same cost for one lane as for 64: with RT_MERGED_T_MIN > 1 the
            // rays that wait for a test sit out rounds until that many wait, or until no ray wants a node step.
            if (RT_MERGED_T_MIN > 1 && (uint32_t)__popcll(__ballot(wantTest)) < RT_MERGED_T_MIN && __ballot(wantNode || wantIdx) != 0ull) wantTest = false;
            const bool moreInLeaf = wantTest && leafI + 1 < leafN;
            const uint2 nd = tb.nodes[wantNode ? node : 0u];
            const int idx = tb.leafIdx[wantIdx ? leafBase + (uint32_t)leafI : (moreInLeaf ? leafBase + (uint32_t)leafI + 1u : 0u)];
            const uint32_t rec = wantTest ? 3u * (uint32_t)prim : 0u;
            const float4 q0 = tb.triHot[rec + 0], q1 = tb.triHot[rec + 1], q2 = tb.triHot[rec + 2];
            __builtin_amdgcn_sched_barrier(0);   // the loads above are issued before anything below waits for one of them
            if (wantNode) {
                if ((nd.y & 3u) != 3u) {
                    if (STATS) ++cn.interior;
                    const int axis = (int)(nd.y & 3u);
                    const float split = __uint_as_float(nd.x);
                    const float oa = comp3(r.o, axis), da = comp3(r.d, axis), ia = comp3(r.invDir, axis);
                    const float tPlane = (split - oa) * ia;
                    const bool belowFirst = (oa < split) || (oa == split && da <= 0);
                    const uint32_t above = nd.y >> 2;
                    const uint32_t first = belowFirst ? node + 1 : above;
                    const uint32_t second = belowFirst ? above : node + 1;
                    if (tPlane > tMax || tPlane <= 0) {
                        node = first;
                    } else if (tPlane < tMin) {
                        node = second;
                    } else {
                        if (sp < SL) {
                            ldsStack[(sp * 2 + 0) * RT_BLOCK + tid] = second;
                            ldsStack[(sp * 2 + 1) * RT_BLOCK + tid] = __float_as_uint(tMax);
                        } else if (sp < RT_STACK_TOTAL) {
                            const int q = sp - SL;
                            privNode[q] = second;
                            privTMax[q] = tMax;
                        }
                        sp = sp < RT_STACK_TOTAL ? sp + 1 : sp;   // see kd_trace: never taken for an accepted tree
                        node = first;
                        tMax = tPlane;
                    }
                } else {
                    if (STATS) ++cn.leaf;
                    const int nP = (int)(nd.y >> 2);
                    if (nP > 0) {
                        leafN = nP;
                        leafI = 0;
                        leafBase = nd.x;
                        prim = nP == 1 ? (int)nd.x : -1;   // a leaf of one primitive names it; a longer one names its index range
                    } else if (!pop()) {   // empty leaf
                        job.finish(jobId, hit);
                        active = false;
                    }
                }
            }
            if (wantIdx) prim = idx;
            if (wantTest) {
                if (STATS) {
                    ++cn.tri;
                    if (leafN > 1) ++cn.leafIdx;
                }
                float t, b0, b1, b2;
                ++leafI;
                bool accepted;
                if (SPH && q0.w == RT_PRIM_SPHERE) {
                    f3 oo, dd;
                    accepted = sphere_test(sc.spheres[__float_as_int(q0.x)], r.o, r.d, rayTMax, oo, dd, t);
                    b0 = t;
                    b1 = b2 = 0.f;
                } else {
                    accepted = tri_test<PERM>(r, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z), rayTMax, t, b0, b1, b2);
                }
                if (accepted) {
                    if (!CLOSEST) {
                        hit.prim = prim;
                        job.finish(jobId, hit);
                        active = false;
                    } else if (q0.w != RT_PRIM_DEGENERATE) {
                        rayTMax = t;
                        hit.prim = prim;
                        hit.t = t;
                        hit.b0 = b0; hit.b1 = b1; hit.b2 = b2;
                    }
                }
                prim = idx;   // the leaf's next primitive, if it has one
                if (active && leafI == leafN) {
                    leafI = leafN = 0;
                    if (!pop()) {
                        job.finish(jobId, hit);
                        active = false;
                    }
                }
            }
        } else {
        // descend: one node per lane and round; more rounds while most of the wave is still between leaves
        for (;;) {
            if (active && leafI >= leafN) {
                {
                    const uint2 nd = tb.nodes[node];
                    if ((nd.y & 3u) != 3u) {
                        if (STATS) ++cn.interior;
                        const int axis = (int)(nd.y & 3u);
                        const float split = __uint_as_float(nd.x);
                        const float oa = comp3(r.o, axis), da = comp3(r.d, axis), ia = comp3(r.invDir, axis);
                        const float tPlane = (split - oa) * ia;
                        const bool belowFirst = (oa < split) || (oa == split && da <= 0);
                        const uint32_t above = nd.y >> 2;
                        const uint32_t first = belowFirst ? node + 1 : above;
                        const uint32_t second = belowFirst ? above : node + 1;
                        if (tPlane > tMax || tPlane <= 0) {
                            node = first;
                        } else if (tPlane < tMin) {
                            node = second;
                        } else {
                            if (sp < SL) {
                                ldsStack[(sp * 2 + 0) * RT_BLOCK + tid] = second;
                                ldsStack[(sp * 2 + 1) * RT_BLOCK + tid] = __float_as_uint(tMax);
                            } else if (sp < RT_STACK_TOTAL) {
                                const int q = sp - SL;
                                privNode[q] = second;
                                privTMax[q] = tMax;
                            }
                            sp = sp < RT_STACK_TOTAL ? sp + 1 : sp;   // see kd_trace: never taken for an accepted tree
                            node = first;
                            tMax = tPlane;
                        }
                    } else {
                        if (STATS) ++cn.leaf;
                        const int nP = (int)(nd.y >> 2);
                        if (nP > 0) {
                            leafN = nP;
                            leafI = 0;
                            leafBase = nd.x;
                        } else if (!pop()) {   // empty leaf
                            job.finish(jobId, hit);
                            active = false;
                        }
                    }
                }
            }
            if ((uint32_t)__popcll(__ballot(active && leafI >= leafN)) < (PERM ? RT_D_MAJ_LDS : RT_D_MAJ)) break;   // PERM: tables in LDS
        }
        // test: one primitive per lane and round; more rounds while most of the wave is inside leaves
        for (;;) {
            if (active && leafI < leafN) {
                int prim;
                if (leafN == 1) prim = (int)leafBase;
                else {
                    prim = tb.leafIdx[leafBase + (uint32_t)leafI];
                    if (STATS) ++cn.leafIdx;
                }
                if (STATS) ++cn.tri;
                const uint32_t rec = tri_record<PERM>(triBase, prim);
                const float4 q0 = tb.triHot[rec + 0];
                const float4 q1 = tb.triHot[rec + 1];
                const float4 q2 = tb.triHot[rec + 2];
                float t, b0, b1, b2;
                ++leafI;
                bool accepted;
                if (SPH && q0.w == RT_PRIM_SPHERE) {
                    f3 oo, dd;
                    accepted = sphere_test(sc.spheres[__float_as_int(q0.x)], r.o, r.d, rayTMax, oo, dd, t);
                    b0 = t;
                    b1 = b2 = 0.f;
                } else {
                    accepted = tri_test<PERM>(r, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z), rayTMax, t, b0, b1, b2);
                }
                if (accepted) {
                    if (!CLOSEST) {
                        hit.prim = prim;
                        job.finish(jobId, hit);
                        active = false;
                    } else if (q0.w != RT_PRIM_DEGENERATE) {
                        rayTMax = t;
                        hit.prim = prim;
                        hit.t = t;
                        hit.b0 = b0; hit.b1 = b1; hit.b2 = b2;
                    }
                }
                if (active && leafI == leafN) {
                    leafI = leafN = 0;
                    if (!pop()) {
                        job.finish(jobId, hit);
                        active = false;
                    }
                }
            }
            if ((uint32_t)__popcll(__ballot(active && leafI < leafN)) < (PERM ? RT_T_MAJ_LDS : RT_T_MAJ)) break;
        }
        }
    }
}

// Jobs of the batch entry points (rt_trace_closest / rt_trace_any): rays from three arrays, results to the caller's arrays.
struct TraceClosestJob {
    const float *o, *d, *tmax;
    int *prim;
    float *tOut, *bary;
    RT_DEV bool fetch(uint32_t j, f3 &ro, f3 &rd, float &tm) const {
        const size_t i = j;
        ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        tm = tmax ? tmax[i] : RT_INF;
        return true;
    }
    RT_DEV void finish(uint32_t j, const HitRec &h) const {
        const size_t i = j;
        if (prim) prim[i] = h.prim;
        if (tOut) tOut[i] = h.t;
        if (bary) { bary[3 * i] = h.b0; bary[3 * i + 1] = h.b1; bary[3 * i + 2] = h.b2; }
    }
};

struct TraceAnyJob {
    const float *o, *d, *tmax;
    uint8_t *occluded;
    RT_DEV bool fetch(uint32_t j, f3 &ro, f3 &rd, float &tm) const {
        const size_t i = j;
        ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        tm = tmax ? tmax[i] : RT_INF;
        return true;
    }
    RT_DEV void finish(uint32_t j, const HitRec &h) const { occluded[j] = h.prim >= 0 ? 1 : 0; }
};
