// KD-tree traversal and the watertight ray/triangle test on the device.
//
// One lane owns one ray and performs exactly the sequence of node visits, primitive tests and float
// operations of the reference's KdTree::hit pair (reference src/core/Accelerators/KDTree.cpp:284-490,
// root clip src/core/Utils/Math.h:572-601) and ATriangleShape::hit pair
// (src/core/Shape/TriangleShape.cpp:178-385): same comparisons (so NaNs from 0*inf fall through the
// same branches), same visit order (ties in t let the last tested primitive win), no mailboxing.
// What changes is the schedule: lanes descend interior nodes together and meet at leaves, the
// per-ray shear constants are computed once per ray instead of once per triangle, and the todo stack
// lives in LDS (first STACK_LDS entries) with a private-memory tail.
#pragma once
#include "dmath.h"

#define RT_BLOCK 256   // threads per workgroup of every kernel: four waves (the block-wide appends and rings count on exactly four)
#define RT_STACK_TOTAL 64     // the reference's todo[64]
// todo entries per lane kept in LDS; the rest lives in private memory. Small (LDS-resident) scenes have
// shallow trees, so a short LDS part keeps more workgroups resident per CU.
#ifndef RT_STACK_LDS_SMALL
#define RT_STACK_LDS_SMALL 4
#endif
#ifndef RT_STACK_LDS_LARGE
#define RT_STACK_LDS_LARGE 12
#endif
// Empty statement that the compiler must assume rewrites two vector registers: placed between two reads of a value, it
// keeps the compiler from merging them into one read through a pointer.
#ifndef RT_PIN_VGPRS
#define RT_PIN_VGPRS(a, b) asm volatile("" : "+v"(a), "+v"(b))
#endif
template <bool LDS_SCENE> struct StackCfg { static constexpr int LDS_N = LDS_SCENE ? RT_STACK_LDS_SMALL : RT_STACK_LDS_LARGE; };

struct DevMaterial {
    float r, g, b, eta;
    int type, pad0, pad1, pad2;
};
struct DevLight {
    float er, eg, eb;
    int twoSided;
    int prim;
    float area;
    float pad0, pad1;
};

// ASphereShape: the entity's transform pair (column-major) and the radius. A sphere primitive's triHot record is
// {bits(sphere index) x3, 2.0f}: the w of p0 tells the traversal which test to run.
struct DevSphere {
    float w2o[16], o2w[16];
    float radius, pad0, pad1, pad2;
};
#define RT_PRIM_DEGENERATE 1.f
#define RT_PRIM_SPHERE 2.f

// Device scene: separate tables per access pattern. Records are 16-byte multiples so a lane's gather
// is one or a few dwordx4 loads.
//   nodes     uint2  per node   {split | onePrim | indexOffset, axis/leaf flag | child or count << 2}
//   leafIdx   int    per leaf reference
//   triHot    float4 x3 per primitive: p0, p1, p2 (w of p0 = 1.0f when the triangle is degenerate for
//             the closest-hit flavour, TriangleShape.cpp:414-421). Scenes small enough for LDS carry three
//             copies, one per axis permutation of the triangle test (block (kz + 1) % 3 holds the
//             coordinates permuted for kz; block 0 is the identity), so the test does not permute per ray.
//   triShade  float4 x5 per primitive: {n0, ss.x} {n1, ss.y} {n2, ss.z} {geometric n, hasNormals}
//             {materialId, lightId, -, -} — read by shading only
struct DevScene {
    const uint2 *nodes;
    const int *leafIdx;
    const float4 *triHot;
    const float4 *triShade;
    const DevMaterial *mats;
    const DevLight *lights;
    const float *lightCdf;    // nLights + 1
    const float *lightFunc;   // nLights
    const DevSphere *spheres;
    float bmin[3], bmax[3];
    uint32_t nNodes, nLeafIdx, nPrims, nMats, nLights, nSpheres;
    float lightFuncInt;
    uint32_t ldsSceneBytes;   // nodes + leafIdx + triHot, each padded to 16 B; 0 = does not fit in LDS
    uint32_t triCopies;       // 3 when triHot holds the permuted copies, else 1
};

struct RayCtx {
    f3 o, d, invDir;
    f3 op;             // origin with the triangle test's axis permutation applied
    int kz;
    float Sx, Sy, Sz;
};

struct TravCounters {
    uint32_t interior, leaf, leafIdx, tri;
    uint32_t rays;   // filled by trace_stream only
};

RT_DEV f3 perm3(f3 v, int kz) {
    // kz = 0: (y, z, x); kz = 1: (z, x, y); kz = 2: (x, y, z)   (kx = kz+1 mod 3, ky = kx+1 mod 3)
    return mk3(kz == 0 ? v.y : (kz == 1 ? v.z : v.x), kz == 0 ? v.z : (kz == 1 ? v.x : v.y), kz == 0 ? v.x : (kz == 1 ? v.y : v.z));
}

RT_DEV void ray_setup(RayCtx &r, f3 o, f3 d) {
    r.o = o;
    r.d = d;
    r.invDir = mk3(1 / d.x, 1 / d.y, 1 / d.z);
    f3 a = mk3(fabsf(d.x), fabsf(d.y), fabsf(d.z));
    r.kz = (a.x > a.y) ? ((a.x > a.z) ? 0 : 2) : ((a.y > a.z) ? 1 : 2);
    f3 dp = perm3(d, r.kz);
    r.Sx = -dp.x / dp.z;
    r.Sy = -dp.y / dp.z;
    r.Sz = comp3(r.invDir, r.kz);   // 1.f / dp.z, and dp.z is d's component kz: the quotient already computed for invDir
    r.op = perm3(o, r.kz);
}

// Root slab clip (BBox3::hit).
RT_DEV bool bounds_clip(const DevScene &sc, const RayCtx &r, float rayTMax, float &hitt0, float &hitt1) {
    float t0 = 0, t1 = rayTMax;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float inv = comp3(r.invDir, i);
        const float oi = comp3(r.o, i);
        float tNear = (sc.bmin[i] - oi) * inv;
        float tFar = (sc.bmax[i] - oi) * inv;
        if (tNear > tFar) { float tmp = tNear; tNear = tFar; tFar = tmp; }
        tFar *= 1 + 2 * gamma_err(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    hitt0 = t0;
    hitt1 = t1;
    return true;
}

// Ray/triangle test. rayTMax is the ray's current m_tMax. PERM: the vertices come from the copy of the
// table that is already permuted for this ray's kz.
// Two floats that the compiler keeps in a register pair: +, - and * on them are the packed FP32 instructions (two IEEE
// operations per issue slot, each rounded exactly as the scalar one).
typedef float v2f __attribute__((vector_size(8)));
RT_HD v2f mk2(float a, float b) { v2f r = {a, b}; return r; }

template <bool PERM>
RT_DEV bool tri_test(const RayCtx &r, f3 p0, f3 p1, f3 p2, float rayTMax, float &tOut, float &b0o, float &b1o, float &b2o) {
    const f3 q0 = PERM ? p0 : perm3(p0, r.kz), q1 = PERM ? p1 : perm3(p1, r.kz), q2 = PERM ? p2 : perm3(p2, r.kz);
    // translate, shear, edge functions: x and y of a vertex travel as a pair. Every product and every sum is rounded once,
    // in the order of the reference's scalar code (p.x += Sx * p.z; e0 = p1.x * p2.y - p1.y * p2.x ...).
    const v2f oxy = mk2(r.op.x, r.op.y), sxy = mk2(r.Sx, r.Sy);
    v2f a0 = mk2(q0.x, q0.y) - oxy, a1 = mk2(q1.x, q1.y) - oxy, a2 = mk2(q2.x, q2.y) - oxy;
    float z0 = q0.z - r.op.z, z1 = q1.z - r.op.z, z2 = q2.z - r.op.z;
    a0 = a0 + sxy * mk2(z0, z0);
    a1 = a1 + sxy * mk2(z1, z1);
    a2 = a2 + sxy * mk2(z2, z2);
    const v2f m0 = a1 * mk2(a2[1], a2[0]), m1 = a2 * mk2(a0[1], a0[0]), m2 = a0 * mk2(a1[1], a1[0]);
    float e0 = m0[0] - m0[1];
    float e1 = m1[0] - m1[1];
    float e2 = m2[0] - m2[1];
    RT_PIN_VGPRS(e0, e1);   // three plain subtractions: pairing two of them would cost three register moves
    // e0 == 0 || e1 == 0 || e2 == 0 (a NaN is not zero, and fminf skips it)
    if (fminf(fminf(fabsf(e0), fabsf(e1)), fabsf(e2)) == 0.0f) {
        double p2txp1ty = (double)a2[0] * (double)a1[1];
        double p2typ1tx = (double)a2[1] * (double)a1[0];
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)a0[0] * (double)a2[1];
        double p0typ2tx = (double)a0[1] * (double)a2[0];
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)a1[0] * (double)a0[1];
        double p1typ0tx = (double)a1[1] * (double)a0[0];
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    // (e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0): the smallest is negative and the largest positive
    // (a NaN compares false in the reference's form; fminf / fmaxf skip it, and three NaNs give a NaN, which compares false)
    const float eMin = fminf(fminf(e0, e1), e2), eMax = fmaxf(fmaxf(e0, e1), e2);
    if ((eMin < 0) & (eMax > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    z0 *= r.Sz;
    z1 *= r.Sz;
    z2 *= r.Sz;
    float tScaled = e0 * z0 + e1 * z1 + e2 * z2;
    if (det < 0 && (tScaled >= 0 || tScaled < rayTMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > rayTMax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = gmax(fabsf(z0), gmax(fabsf(z1), fabsf(z2)));
    float deltaZ = gamma_err(3) * maxZt;
    float maxXt = gmax(fabsf(a0[0]), gmax(fabsf(a1[0]), fabsf(a2[0])));
    float maxYt = gmax(fabsf(a0[1]), gmax(fabsf(a1[1]), fabsf(a2[1])));
    float deltaX = gamma_err(5) * (maxXt + maxZt);
    float deltaY = gamma_err(5) * (maxYt + maxZt);
    float deltaE = 2 * (gamma_err(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = gmax(fabsf(e0), gmax(fabsf(e1), fabsf(e2)));
    float deltaT = 3 * (gamma_err(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
    if (t <= deltaT) return false;
    tOut = t;
    b0o = b0; b1o = b1; b2o = b2;
    return true;
}

// Transform::operator()(Vec3, w) (Utils/Transform.h:75-88); glm's mat4 * vec4 sums (c0 x + c1 y) + (c2 z + c3 w).
RT_DEV f3 xf_apply(const float *m, f3 p, float w) {
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = (m[0 * 4 + k] * p.x + m[1 * 4 + k] * p.y) + (m[2 * 4 + k] * p.z + m[3 * 4 + k] * w);
    if (w == 0.f) return mk3(r[0], r[1], r[2]);
    if (r[3] == 1) return mk3(r[0], r[1], r[2]);
    return mk3(r[0], r[1], r[2]) / r[3];
}

// The quadratic of ASphereShape::hit, both flavours (Shape/SphereShape.cpp:129-200). The ray goes to object space
// through Transform::operator()(Ray) (Utils/Transform.h:90-97), whose Ray constructor normalises the direction again.
RT_DEV bool sphere_test(const DevSphere &sp, f3 o, f3 d, float rayTMax, f3 &oObj, f3 &dObj, float &tHit) {
    oObj = xf_apply(sp.w2o, o, 1.0f);
    dObj = normalize3(xf_apply(sp.w2o, d, 0.0f));
    const float a = dot3(dObj, dObj);
    const float b = dot3(oObj, dObj);
    const float c = dot3(oObj, oObj) - sp.radius * sp.radius;
    const float discriminant = b * b - a * c;
    if (discriminant <= 0) return false;
    const float sq = sqrtf(discriminant);
    float t0 = (-b - sq) / a;
    float t1 = (-b + sq) / a;
    if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > rayTMax || t1 <= 0) return false;
    float tShapeHit = t0;
    if (tShapeHit <= 0) {
        tShapeHit = t1;
        if (tShapeHit > rayTMax) return false;
    }
    tHit = tShapeHit;
    return true;
}

// Result of a traversal. For a triangle b0..b2 are the barycentrics; for a sphere b0 repeats t and b1 = b2 = 0
// (the surface of a sphere hit is rebuilt from the ray and t).
struct HitRec {
    int prim;
    float t, b0, b1, b2;
};

// Pointers to the three traversal tables, either in HBM or staged in LDS.
struct SceneTables {
    const uint2 *nodes;
    const int *leafIdx;
    const float4 *triHot;
};

// Todo stack. The reference's entries are (node, tMin, tMax) (Accelerators/KDTree.h:48-52). tMin is not stored
// here: an entry's tMin is the split-plane distance that became the current tMax when the entry was pushed, and
// every push/pop pair below it restores tMax, so at the moment the entry is popped the current tMax holds exactly
// that value. Entry e of lane `tid` lives at stk[(e * 2 + k) * RT_BLOCK + tid] (k = 0 node, 1 tMax) so that lanes at
// the same depth hit consecutive banks.
// First float4 of the triangle table copy that matches the ray's permutation.
template <bool PERM> RT_DEV uint32_t tri_table_base(const DevScene &sc, int kz) {
    return PERM ? (uint32_t)(kz == 2 ? 0 : kz + 1) * sc.nPrims * 3u : 0u;
}

// Index of a primitive's first float4 in the triangle table. Tables in LDS (PERM; a few hundred primitives at most): one
// 24-bit multiply-add; written as 3 * prim the compiler forms a 64-bit multiply-add per lane, four issue slots.
template <bool PERM> RT_DEV uint32_t tri_record(uint32_t triBase, int prim) {
    return PERM ? RT_MUL24((uint32_t)prim, 3u) + triBase : 3u * (uint32_t)prim + triBase;
}

// SPH: the scene has sphere primitives (scenes without them compile the sphere test away).
template <bool CLOSEST, bool STATS, int SL, bool PERM, bool SPH>
RT_DEV bool kd_trace(const DevScene &sc, const SceneTables &tb, uint32_t *ldsStack, uint32_t tid, const RayCtx &r, float rayTMaxIn,
                     HitRec &hit, TravCounters &cn) {
    float tMin, tMax;
    float rayTMax = rayTMaxIn;
    hit.prim = -1;
    hit.t = rayTMaxIn;
    hit.b0 = hit.b1 = hit.b2 = 0.f;
    if (!bounds_clip(sc, r, rayTMax, tMin, tMax)) return false;

    uint32_t privNode[RT_STACK_TOTAL - SL];
    float privTMax[RT_STACK_TOTAL - SL];
    int sp = 0;
    uint32_t node = 0;
    bool found = false;
    const uint32_t triBase = tri_table_base<PERM>(sc, r.kz);
    for (;;) {
        if (CLOSEST && rayTMax < tMin) break;
        uint2 nd = tb.nodes[node];
        while ((nd.y & 3u) != 3u) {
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
                // A tree that rt_upload_scene accepted never needs more than RT_STACK_TOTAL entries; should the tables be
                // damaged in memory, the entry is dropped rather than written outside the stack.
                sp = sp < RT_STACK_TOTAL ? sp + 1 : sp;
                node = first;
                tMax = tPlane;
            }
            nd = tb.nodes[node];
        }
        // leaf
        if (STATS) ++cn.leaf;
        const int n = (int)(nd.y >> 2);
        for (int i = 0; i < n; ++i) {
            int prim;
            if (n == 1) prim = (int)nd.x;
            else {
                prim = tb.leafIdx[nd.x + i];
                if (STATS) ++cn.leafIdx;
            }
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
                if (!CLOSEST) { hit.prim = prim; return true; }
                if (q0.w != RT_PRIM_DEGENERATE) {
                    rayTMax = t;
                    hit.prim = prim;
                    hit.t = t;
                    hit.b0 = b0; hit.b1 = b1; hit.b2 = b2;
                    found = true;
                }
            }
        }
        if (sp == 0) break;
        --sp;
        tMin = tMax;   // the segment after the one just finished starts where that one ended
        // LDS part read unconditionally (clamped index), private part under its own branch: see trace_stream's pop
        const int e = sp < SL ? sp : SL - 1;
        node = ldsStack[(e * 2 + 0) * RT_BLOCK + tid];
        tMax = __uint_as_float(ldsStack[(e * 2 + 1) * RT_BLOCK + tid]);
        RT_PIN_VGPRS(node, tMax);   // keeps the two LDS reads as LDS reads (no merge with the branch below)
        if (sp >= SL) {
            const int q = sp - SL;
            node = privNode[q];
            tMax = privTMax[q];
        }
    }
    return found;
}

// Cooperative copy of the traversal tables into LDS. `smem` must hold sc.ldsSceneBytes bytes, 16-B aligned.
RT_DEV void stage_scene_lds(const DevScene &sc, uint4 *smem, SceneTables &tb) {
    const uint32_t nodeVec = (sc.nNodes * 8 + 15) / 16;
    const uint32_t idxVec = (sc.nLeafIdx * 4 + 15) / 16;
    const uint32_t triVec = sc.nPrims * 3 * sc.triCopies;
    const uint4 *gn = reinterpret_cast<const uint4 *>(sc.nodes);
    const uint4 *gi = reinterpret_cast<const uint4 *>(sc.leafIdx);
    const uint4 *gt = reinterpret_cast<const uint4 *>(sc.triHot);
    for (uint32_t i = threadIdx.x; i < nodeVec; i += blockDim.x) smem[i] = gn[i];
    for (uint32_t i = threadIdx.x; i < idxVec; i += blockDim.x) smem[nodeVec + i] = gi[i];
    for (uint32_t i = threadIdx.x; i < triVec; i += blockDim.x) smem[nodeVec + idxVec + i] = gt[i];
    tb.nodes = reinterpret_cast<const uint2 *>(smem);
    tb.leafIdx = reinterpret_cast<const int *>(smem + nodeVec);
    tb.triHot = reinterpret_cast<const float4 *>(smem + nodeVec + idxVec);
    __syncthreads();
}
