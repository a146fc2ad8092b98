// Work-queue KD-tree tracer: rays are not pinned to lanes.
//
// kd_trace (dtraverse.h) and trace_stream (dstream.h) keep a ray in the registers of one lane from its first node to
// its last, so every vector instruction of a node step runs for the lanes that happen to be between leaves and every
// instruction of a triangle test for the lanes that happen to be inside one: the hardware counters show 12 (lockstep)
// to 19 (streaming) of 64 lanes doing useful work per instruction. Here a wave keeps the state of R > 64 rays in LDS
// and three queues of ray slots: rays that need a node step (D), rays that need a primitive test (T), free slots (F).
// A round takes up to 64 slots from ONE queue, so all of its lanes run the same code: load the slot's state, do the
// step, store the state, append the slot to the queue of its next step. A round of node steps is a round of 64 node
// fetches in flight, whatever the rays' histories.
//
// Per ray the node visits, primitive tests and float operations are exactly those of kd_trace, hence of the
// reference's KdTree::hit pair (src/core/Accelerators/KDTree.cpp:284-490) and ATriangleShape::hit pair
// (src/core/Shape/TriangleShape.cpp:178-385): a ray's steps are executed one after the other (a slot is in one queue at
// a time), only by varying lanes. What changes is the order in which DIFFERENT rays make progress.
//
// LDS of one wave (dwords; field f of slot s lives at f * R + s, so lanes that hold different slots hit different
// banks unless the slots are congruent mod 64):
//   [ray fields: QF_COUNT x R][todo stack: SL x {node, tMax} x R][queue D: R][queue T: R][queue F: R]
// Todo entries beyond the first SL of a ray go to a buffer in global memory (one region per resident wave).
//
// The Job type is the one of trace_stream:
//   bool fetch(uint32_t j, f3 &o, f3 &d, float &tMax)   ray j of the batch (false: entry j has no ray)
//   void finish(uint32_t j, const HitRec &h)            called exactly once per fetched ray; h.prim < 0 is a miss
#pragma once
#include "dtraverse.h"

#ifndef RT_Q_SLOTS
#define RT_Q_SLOTS 128        // ray slots per wave
#endif
#ifndef RT_Q_STACK_LDS
#define RT_Q_STACK_LDS 6      // todo entries per ray kept in LDS
#endif
#ifndef RT_Q_REFILL_MIN
#define RT_Q_REFILL_MIN 48    // free slots needed before the wave fetches new rays
#endif

enum {
    QF_OX = 0, QF_OY, QF_OZ, QF_DX, QF_DY, QF_DZ, QF_IX, QF_IY, QF_IZ, QF_SX, QF_SY, QF_SZ,   // set when the ray is fetched
    QF_NODE, QF_TMIN, QF_TMAX, QF_RAYTMAX, QF_SP, QF_CUR, QF_END, QF_PRIM, QF_B0, QF_B1, QF_B2, QF_JOB,
    QF_COUNT
};
#define RT_Q_SINGLE 0x80000000u   // QF_CUR of a one-primitive leaf holds the primitive id itself (KdTreeNode::m_oneHitable)

template <int R, int SL> struct QueueLds {
    static constexpr uint32_t kStack = QF_COUNT * R;
    static constexpr uint32_t kQD = kStack + SL * 2 * R;
    static constexpr uint32_t kQT = kQD + R;
    static constexpr uint32_t kQF = kQT + R;
    static constexpr uint32_t kDwordsPerWave = kQF + R;
    static constexpr uint32_t kBytesPerBlock = kDwordsPerWave * 4 * (RT_BLOCK / 64);
    static constexpr uint32_t kOverflowPerWave = (uint32_t)R * (RT_STACK_TOTAL - SL);   // uint2 entries in global memory
};

// Makes the wave's LDS writes visible to its other lanes' later reads, in program order.
RT_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool CLOSEST, bool STATS, bool PERM, bool SPH, int R, int SL, class Job>
RT_DEV void trace_queue(const DevScene &sc, const SceneTables &tb, uint32_t *lds, uint2 *overflow, uint32_t n, Job &job, TravCounters &cn,
                        unsigned long long *guard) {
    using L = QueueLds<R, SL>;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    float *ldsf = reinterpret_cast<float *>(lds);
    uint32_t *qD = lds + L::kQD, *qT = lds + L::kQT, *qF = lds + L::kQF;
    uint32_t *stk = lds + L::kStack;
    // Integrity guard: every index read back from the slot state is checked before it addresses memory. A violation is an
    // internal error (it cannot come from the caller's data, which rt_upload_scene validates): it is recorded in *guard
    // and the ray ends as a miss instead of touching memory it does not own.
    auto bad = [&](bool violated, unsigned long long code) -> bool {
        if (violated) atomicOr(guard, code);
        return violated;
    };

    // this wave's share of the batch: a contiguous range, so that a refill reads consecutive rays
    const uint32_t nWaves = gridDim.x * (RT_BLOCK / 64);
    const uint32_t waveId = blockIdx.x * (RT_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t per = (uint32_t)(((uint64_t)n + nWaves - 1) / nWaves);
    const uint64_t first64 = (uint64_t)waveId * per, last64 = first64 + per;
    uint32_t next = first64 < (uint64_t)n ? (uint32_t)first64 : n;
    const uint32_t last = last64 < (uint64_t)n ? (uint32_t)last64 : n;

    // queues: rings of R entries; heads and counts are the same in every lane
    uint32_t hD = 0, nD = 0, hT = 0, nT = 0, hF = 0, nF = R;
    for (uint32_t s = lane; s < (uint32_t)R; s += 64u) qF[s] = s;
    wave_sync();

    auto fld = [&](int f, uint32_t slot) -> uint32_t & { return lds[(uint32_t)f * R + slot]; };
    auto fldf = [&](int f, uint32_t slot) -> float & { return ldsf[(uint32_t)f * R + slot]; };
    auto push = [&](uint32_t *q, uint32_t head, uint32_t &count, bool want, uint32_t slot) {
        const unsigned long long m = __ballot(want);
        if (want) q[(head + count + (uint32_t)__popcll(m & below)) % (uint32_t)R] = slot;
        count += (uint32_t)__popcll(m);
    };
    auto finish = [&](uint32_t slot) {
        HitRec h;
        h.prim = (int)fld(QF_PRIM, slot);
        h.t = fldf(QF_RAYTMAX, slot);
        h.b0 = fldf(QF_B0, slot);
        h.b1 = fldf(QF_B1, slot);
        h.b2 = fldf(QF_B2, slot);
        const uint32_t j = fld(QF_JOB, slot);
        if (!bad(j >= n, 2ull)) job.finish(j, h);
    };
    // Next todo entry of the ray in `slot`, whose current segment ended at tMaxNow. False: the ray is finished (stack
    // empty, or the closest hit so far lies before the entry).
    auto pop = [&](uint32_t slot, float tMaxNow) -> bool {
        uint32_t sp = fld(QF_SP, slot);
        if (sp == 0) return false;
        if (bad(sp > (uint32_t)RT_STACK_TOTAL, 1ull)) return false;
        --sp;
        uint32_t node;
        float tMax;
        if (sp < (uint32_t)SL) {
            node = stk[(sp * 2 + 0) * R + slot];
            tMax = __uint_as_float(stk[(sp * 2 + 1) * R + slot]);
        } else {
            const uint2 e = overflow[(size_t)slot * (RT_STACK_TOTAL - SL) + (sp - SL)];
            node = e.x;
            tMax = __uint_as_float(e.y);
        }
        fld(QF_SP, slot) = sp;
        fld(QF_NODE, slot) = node;
        fldf(QF_TMIN, slot) = tMaxNow;   // the segment after the one just finished starts where that one ended
        fldf(QF_TMAX, slot) = tMax;
        if (CLOSEST && fldf(QF_RAYTMAX, slot) < tMaxNow) return false;
        return true;
    };

    // every round completes at least one step of one ray; a ray has at most 2 nNodes + nLeafIdx steps. The bound below is
    // far above any real round count and only keeps an internal error from spinning for ever.
    unsigned long long roundsLeft = 64ull + (unsigned long long)per * (1ull << 16);
    for (;;) {
        if (roundsLeft-- == 0ull) { atomicOr(guard, 64ull); break; }
        const bool canRefill = next < last && nF >= (uint32_t)RT_Q_REFILL_MIN;
        int what;   // 0 refill, 1 node steps, 2 primitive tests
        if (nT >= 64u) what = 2;
        else if (nD >= 64u) what = 1;
        else if (canRefill) what = 0;
        else if (nD > 0u || nT > 0u) what = nT > nD ? 2 : 1;
        else if (next < last) what = 0;   // nothing resident: every slot is free
        else break;

        if (what == 0) {
            // ---- fetch up to 64 new rays: set-up and root clip at full width --------------------------------------
            const uint32_t take = min(min(nF, 64u), last - next);
            const bool have = lane < take;
            uint32_t slot = 0;
            bool toD = false;
            if (have) {
                slot = qF[(hF + lane) % (uint32_t)R];
                if (bad(slot >= (uint32_t)R, 128ull)) slot = 0;
                const uint32_t j = next + lane;
                f3 o, d;
                float tm;
                bool valid = false;
                if (job.fetch(j, o, d, tm)) {
                    if (STATS) ++cn.rays;
                    RayCtx r;
                    ray_setup(r, o, d);
                    float t0, t1;
                    valid = bounds_clip(sc, r, tm, t0, t1);
                    if (valid && CLOSEST && tm < t0) valid = false;   // KdTree::hit's first test of the loop (KDTree.cpp:406)
                    if (valid) {
                        fldf(QF_OX, slot) = o.x; fldf(QF_OY, slot) = o.y; fldf(QF_OZ, slot) = o.z;
                        fldf(QF_DX, slot) = d.x; fldf(QF_DY, slot) = d.y; fldf(QF_DZ, slot) = d.z;
                        fldf(QF_IX, slot) = r.invDir.x; fldf(QF_IY, slot) = r.invDir.y; fldf(QF_IZ, slot) = r.invDir.z;
                        fldf(QF_SX, slot) = r.Sx; fldf(QF_SY, slot) = r.Sy; fldf(QF_SZ, slot) = r.Sz;
                        fld(QF_NODE, slot) = 0u;
                        fldf(QF_TMIN, slot) = t0;
                        fldf(QF_TMAX, slot) = t1;
                        fldf(QF_RAYTMAX, slot) = tm;
                        fld(QF_SP, slot) = 0u;
                        fld(QF_PRIM, slot) = 0xffffffffu;
                        fldf(QF_B0, slot) = 0.f; fldf(QF_B1, slot) = 0.f; fldf(QF_B2, slot) = 0.f;
                        fld(QF_JOB, slot) = j;
                        toD = true;
                    } else {
                        const HitRec miss = {-1, tm, 0.f, 0.f, 0.f};
                        job.finish(j, miss);
                    }
                }
            }
            next += take;
            // slots of rays that did not enter the tree go back to the end of the free ring
            hF = (hF + take) % (uint32_t)R;
            nF -= take;
            push(qD, hD, nD, toD, slot);
            push(qF, hF, nF, have && !toD, slot);
            wave_sync();
        } else if (what == 1) {
            // ---- one node step for up to 64 rays ------------------------------------------------------------------
            const uint32_t take = min(nD, 64u);
            const bool have = lane < take;
            uint32_t slot = 0;
            bool toD = false, toT = false, fin = false;
            if (have) {
                slot = qD[(hD + lane) % (uint32_t)R];
                if (bad(slot >= (uint32_t)R, 128ull)) slot = 0;
                const uint32_t node = fld(QF_NODE, slot);
                const uint2 nd = bad(node >= sc.nNodes, 4ull) ? make_uint2(0u, 3u) : tb.nodes[node];   // guard: an empty leaf
                if ((nd.y & 3u) != 3u) {
                    if (STATS) ++cn.interior;
                    const uint32_t axis = nd.y & 3u;
                    const float split = __uint_as_float(nd.x);
                    const float oa = fldf(QF_OX + (int)axis, slot), ia = fldf(QF_IX + (int)axis, slot);
                    const float tMin = fldf(QF_TMIN, slot), tMax = fldf(QF_TMAX, slot);
                    const float tPlane = (split - oa) * ia;
                    bool belowFirst = oa < split;
                    if (!belowFirst && oa == split) belowFirst = fldf(QF_DX + (int)axis, slot) <= 0;
                    const uint32_t above = nd.y >> 2;
                    const uint32_t first = belowFirst ? node + 1 : above;
                    const uint32_t second = belowFirst ? above : node + 1;
                    if (tPlane > tMax || tPlane <= 0) {
                        fld(QF_NODE, slot) = first;
                    } else if (tPlane < tMin) {
                        fld(QF_NODE, slot) = second;
                    } else {
                        const uint32_t sp = fld(QF_SP, slot);
                        if (bad(sp >= (uint32_t)RT_STACK_TOTAL, 8ull)) {
                            // guard: drop the entry
                        } else if (sp < (uint32_t)SL) {
                            stk[(sp * 2 + 0) * R + slot] = second;
                            stk[(sp * 2 + 1) * R + slot] = __float_as_uint(tMax);
                        } else {
                            overflow[(size_t)slot * (RT_STACK_TOTAL - SL) + (sp - SL)] = make_uint2(second, __float_as_uint(tMax));
                        }
                        fld(QF_SP, slot) = sp + 1;
                        fld(QF_NODE, slot) = first;
                        fldf(QF_TMAX, slot) = tPlane;
                    }
                    toD = true;
                } else {
                    if (STATS) ++cn.leaf;
                    const uint32_t nP = nd.y >> 2;
                    if (nP == 1u) {
                        fld(QF_CUR, slot) = nd.x | RT_Q_SINGLE;
                        fld(QF_END, slot) = (nd.x | RT_Q_SINGLE) + 1u;
                        toT = true;
                    } else if (nP > 1u) {
                        fld(QF_CUR, slot) = nd.x;
                        fld(QF_END, slot) = nd.x + nP;
                        toT = true;
                    } else if (pop(slot, fldf(QF_TMAX, slot))) {   // empty leaf
                        toD = true;
                    } else {
                        fin = true;
                    }
                }
                if (fin) finish(slot);
            }
            hD = (hD + take) % (uint32_t)R;
            nD -= take;
            push(qD, hD, nD, toD, slot);
            push(qT, hT, nT, toT, slot);
            push(qF, hF, nF, fin, slot);
            wave_sync();
        } else {
            // ---- one primitive test for up to 64 rays -------------------------------------------------------------
            const uint32_t take = min(nT, 64u);
            const bool have = lane < take;
            uint32_t slot = 0;
            bool toD = false, toT = false, fin = false;
            if (have) {
                slot = qT[(hT + lane) % (uint32_t)R];
                if (bad(slot >= (uint32_t)R, 128ull)) slot = 0;
                uint32_t cur = fld(QF_CUR, slot);
                const uint32_t end = fld(QF_END, slot);
                int prim;
                if (cur & RT_Q_SINGLE) prim = (int)(cur & ~RT_Q_SINGLE);
                else {
                    prim = bad(cur >= sc.nLeafIdx, 16ull) ? 0 : tb.leafIdx[cur];
                    if (STATS) ++cn.leafIdx;
                }
                if (bad((uint32_t)prim >= sc.nPrims, 32ull)) prim = 0;
                if (STATS) ++cn.tri;
                RayCtx r;
                r.o = mk3(fldf(QF_OX, slot), fldf(QF_OY, slot), fldf(QF_OZ, slot));
                r.d = mk3(fldf(QF_DX, slot), fldf(QF_DY, slot), fldf(QF_DZ, slot));
                r.Sx = fldf(QF_SX, slot);
                r.Sy = fldf(QF_SY, slot);
                r.Sz = fldf(QF_SZ, slot);
                const f3 a = mk3(fabsf(r.d.x), fabsf(r.d.y), fabsf(r.d.z));
                r.kz = (a.x > a.y) ? ((a.x > a.z) ? 0 : 2) : ((a.y > a.z) ? 1 : 2);
                r.op = perm3(r.o, r.kz);
                r.invDir = mk3(0.f, 0.f, 0.f);   // not used by the tests
                float rayTMax = fldf(QF_RAYTMAX, slot);
                const uint32_t triBase = tri_table_base<PERM>(sc, r.kz);
                const float4 q0 = tb.triHot[triBase + 3 * prim + 0];
                const float4 q1 = tb.triHot[triBase + 3 * prim + 1];
                const float4 q2 = tb.triHot[triBase + 3 * prim + 2];
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
                        fld(QF_PRIM, slot) = (uint32_t)prim;
                        fin = true;
                    } else if (q0.w != RT_PRIM_DEGENERATE) {
                        fldf(QF_RAYTMAX, slot) = t;
                        fld(QF_PRIM, slot) = (uint32_t)prim;
                        fldf(QF_B0, slot) = b0; fldf(QF_B1, slot) = b1; fldf(QF_B2, slot) = b2;
                    }
                }
                if (!fin) {
                    ++cur;
                    if (cur != end) {
                        fld(QF_CUR, slot) = cur;
                        toT = true;
                    } else if (pop(slot, fldf(QF_TMAX, slot))) {
                        toD = true;
                    } else {
                        fin = true;
                    }
                }
                if (fin) finish(slot);
            }
            hT = (hT + take) % (uint32_t)R;
            nT -= take;
            push(qD, hD, nD, toD, slot);
            push(qT, hT, nT, toT, slot);
            push(qF, hF, nF, fin, slot);
            wave_sync();
        }
    }
}
