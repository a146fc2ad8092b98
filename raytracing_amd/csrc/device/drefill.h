// Refilling KD-tree tracer for scenes whose tables sit in LDS: the lane schedule of trace_stream (dstream.h) without its
// pool of prepared rays and within the register and LDS budget of the lockstep kernels.
//
// The wavefront path tracer on a scene of a few kilobytes is bound by vector-instruction issue: a SIMD issues one wave-level
// instruction every 2.2 - 4.1 cycles (tools/valu_rate.hip), whatever the number of live lanes, so a kernel's cost is the
// number of wave-level instructions it issues. kd_trace
// (dtraverse.h) issues ~90 of them per bounce ray with 11 of 64 lanes live, because a wave runs until its slowest ray ends
// and every lane waits at every leaf for the lane with the biggest one. trace_stream cuts that to ~60, but its prepared-ray
// pool (16 KB of LDS per workgroup) and 112 registers leave room for 4 waves per SIMD, and the instructions it saves are lost
// again to SIMDs that wait. Here a lane whose ray has ended fetches and prepares its next ray itself, as soon as enough lanes
// are idle to pay for running the set-up code once; between refills the wave alternates between node-step rounds and
// primitive-test rounds as trace_stream does. Per ray the node visits, primitive tests and float operations are those of
// kd_trace, hence of the reference's KdTree::hit pair (src/core/Accelerators/KDTree.cpp:284-490).
//
// The Job type is trace_stream's (fetch / finish).
#pragma once
#include "dtraverse.h"

#ifndef RT_RF_D_MAJ
#define RT_RF_D_MAJ 8         // keep stepping through interior nodes while at least this many lanes are between leaves
#endif
#ifndef RT_RF_T_MAJ
#define RT_RF_T_MAJ 24        // keep testing primitives while at least this many lanes have leaf primitives left
#endif
#ifndef RT_RF_D_MAJ_HBM
#define RT_RF_D_MAJ_HBM 16    // tables in HBM: a node step costs a memory round trip however few lanes take it (see dstream.h)
#endif
#ifndef RT_RF_T_MAJ_HBM
#define RT_RF_T_MAJ_HBM 16
#endif
#ifndef RT_RF_REFILL_MIN
#define RT_RF_REFILL_MIN 12   // idle lanes needed before they fetch new rays (the set-up code costs the same for 1 lane as for 64)
#endif

template <bool CLOSEST, bool STATS, int SL, bool PERM, bool SPH, class Job>
RT_DEV void trace_refill(const DevScene &sc, const SceneTables &tb, uint32_t *ldsStack, uint32_t n, Job &job, TravCounters &cn) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned long long below = (1ull << lane) - 1ull;
    // Rays are taken in chunks of 64, chunk c by wave (c mod number of waves). A wave numbers the rays of its chunks
    // 0 .. total-1 ("virtual index" v: chunk w0 + (v / 64) * nWaves, ray v % 64 of it) and hands them out in that order.
    const uint32_t nChunks = (n + 63u) >> 6;
    const uint32_t nWaves = gridDim.x * (RT_BLOCK / 64);
    const uint32_t w0 = blockIdx.x * (RT_BLOCK / 64) + wave;
    const uint32_t total = w0 < nChunks ? ((nChunks - w0 + nWaves - 1u) / nWaves) * 64u : 0u;
    uint32_t cursor = 0;   // next virtual index; the same in every lane

    uint32_t privNode[RT_STACK_TOTAL - SL];
    float privTMax[RT_STACK_TOTAL - SL];

    RayCtx r;
    r.o = r.d = r.invDir = r.op = mk3(0.f, 0.f, 0.f);
    r.kz = 2;
    r.Sx = r.Sy = r.Sz = 0.f;
    float rayTMax = 0.f, tMin = 0.f, tMax = 0.f;
    uint32_t node = 0, jobId = 0, triBase = 0;
    int sp = 0;
    // Where the lane stands: left > 0 primitives of the current leaf still to test, left == 0 between leaves, left < 0 no ray.
    // leafPos is the position of the leaf's next primitive in the index list; a leaf of one primitive names the primitive
    // itself, marked by the top bit.
    int left = -1;
    uint32_t leafPos = 0;
    // closest hit so far; its distance is rayTMax (a ray that found nothing still carries the tMax it came with)
    int hitPrim = -1;
    float hb0 = 0.f, hb1 = 0.f, hb2 = 0.f;

    auto finish = [&]() {
        const HitRec h = {hitPrim, rayTMax, hb0, hb1, hb2};
        job.finish(jobId, h);
        left = -1;
    };
    // next todo entry; false when the ray is finished (stack empty, or the closest hit so far lies before the entry)
    auto pop = [&]() -> bool {
        if (sp == 0) return false;
        --sp;
        tMin = tMax;
        const int e = sp < SL ? sp : SL - 1;
        node = ldsStack[(e * 2 + 0) * RT_BLOCK + tid];
        tMax = __uint_as_float(ldsStack[(e * 2 + 1) * RT_BLOCK + tid]);
        RT_PIN_VGPRS(node, tMax);   // keeps the two LDS reads as LDS reads (see kd_trace)
        if (sp >= SL) {
            const int q = sp - SL;
            node = privNode[q];
            tMax = privTMax[q];
        }
        if (CLOSEST && rayTMax < tMin) return false;
        return true;
    };

    for (;;) {
        const unsigned long long idle = RT_BALLOT(left < 0);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (cursor < total) {
            if (nIdle >= RT_RF_REFILL_MIN) {
                const uint32_t rank = (uint32_t)__popcll(idle & below);
                const uint32_t v = cursor + rank;
                if (left < 0 && v < total) {
                    const uint32_t j = (w0 + (v >> 6) * nWaves) * 64u + (v & 63u);
                    f3 o, d;
                    float ntm = 0.f;
                    if (j < n && job.fetch(j, o, d, ntm)) {
                        if (STATS) ++cn.rays;
                        ray_setup(r, o, d);
                        jobId = j;
                        rayTMax = ntm;
                        hitPrim = -1;
                        hb0 = hb1 = hb2 = 0.f;
                        node = 0;
                        sp = 0;
                        triBase = tri_table_base<PERM>(sc, r.kz);
                        left = 0;
                        if (!bounds_clip(sc, r, ntm, tMin, tMax) || (CLOSEST && rayTMax < tMin)) finish();
                    }
                }
                cursor += nIdle < total - cursor ? nIdle : total - cursor;
            }
        } else if (nIdle == 64u) {
            break;
        }

        // descend: one node per lane and round; more rounds while enough of the wave is still between leaves
        for (;;) {
            if (left == 0) {
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
                        left = nP;
                        leafPos = nP == 1 ? (nd.x | 0x80000000u) : nd.x;
                    } else if (!pop()) {   // empty leaf
                        finish();
                    }
                }
            }
            if ((uint32_t)__popcll(RT_BALLOT(left == 0)) < (PERM ? RT_RF_D_MAJ : RT_RF_D_MAJ_HBM)) break;   // PERM: tables in LDS
        }
        // test: one primitive per lane and round; more rounds while enough of the wave is inside leaves
        for (;;) {
            if (left > 0) {
                int prim;
                if ((int)leafPos < 0) prim = (int)(leafPos & 0x7fffffffu);
                else {
                    prim = tb.leafIdx[leafPos];
                    if (STATS) ++cn.leafIdx;
                }
                ++leafPos;
                --left;
                if (STATS) ++cn.tri;
                const uint32_t rec = tri_record<PERM>(triBase, prim);
                const float4 q0 = tb.triHot[rec + 0];
                const float4 q1 = tb.triHot[rec + 1];
                const float4 q2 = tb.triHot[rec + 2];
                float t, b0, b1, b2;
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
                        hitPrim = prim;
                        finish();
                    } else if (q0.w != RT_PRIM_DEGENERATE) {
                        rayTMax = t;
                        hitPrim = prim;
                        hb0 = b0; hb1 = b1; hb2 = b2;
                    }
                }
                if (left == 0 && !pop()) finish();
            }
            if ((uint32_t)__popcll(RT_BALLOT(left > 0)) < (PERM ? RT_RF_T_MAJ : RT_RF_T_MAJ_HBM)) break;
        }
    }
}
