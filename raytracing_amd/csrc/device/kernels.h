// HIP kernels of the hot path for gfx950: batch closest/any-hit tracing, the wavefront path tracer
// (raygen -> [extend -> shade -> shadow -> probe] per bounce -> film) and the Whitted integrator.
#pragma once
#include "dshade.h"
#include "dstream.h"
#include "drefill.h"

// ---------------------------------------------------------------------------------------------------
// device-side counters (one array per context)
// ---------------------------------------------------------------------------------------------------
enum {
    ST_RAYS_CLOSEST = 0, ST_RAYS_ANY, ST_CAMERA_SAMPLES, ST_INTERIOR, ST_LEAF, ST_LEAFIDX, ST_TRI, ST_COUNTED_RAYS,
    ST_EXTEND_RAYS, ST_COUNT
};

struct CameraDev {
    float r2c[16], c2w[16];
    int width, height;
    int sb[4];   // sample bounds x0,y0,x1,y1
    // film: Box filter radius, cropped pixel bounds x0,y0,x1,y1 (only these pixels receive samples), MaxLum clamp
    float filterRadius[2];
    int cb[4];
    float maxSampleLuminance;
};

// APerspectiveCamera::castingRay (Camera/Camera.cpp:87-95): three normalisations, as the reference's Ray constructor does.
RT_DEV void camera_ray(const CameraDev &cam, float fx, float fy, f3 &o, f3 &d) {
    const f3 pCamera = xf_apply(cam.r2c, mk3(fx, fy, 0.f), 1.0f);
    const f3 dir = normalize3(normalize3(pCamera));
    o = xf_apply(cam.c2w, mk3(0.f, 0.f, 0.f), 1.0f);
    d = normalize3(xf_apply(cam.c2w, dir, 0.0f));
}

// Origin of every camera ray (APerspectiveCamera::castingRay, Camera/Camera.cpp:87-95, starts all rays at the
// camera-space origin): the value camera_ray returns as `o` for any film position.
RT_DEV float4 camera_origin(const CameraDev &cam) {
    f3 o, d;
    camera_ray(cam, 0.f, 0.f, o, d);
    return make_float4(o.x, o.y, o.z, 0.f);
}

RT_DEV void flush_counters(unsigned long long *stats, const TravCounters &cn, uint32_t rays) {
    atomicAdd(&stats[ST_INTERIOR], (unsigned long long)cn.interior);
    atomicAdd(&stats[ST_LEAF], (unsigned long long)cn.leaf);
    atomicAdd(&stats[ST_LEAFIDX], (unsigned long long)cn.leafIdx);
    atomicAdd(&stats[ST_TRI], (unsigned long long)cn.tri);
    atomicAdd(&stats[ST_COUNTED_RAYS], (unsigned long long)rays);
}

// Dynamic LDS layout of every tracing kernel: [todo stack: LDS_N*2*RT_BLOCK dwords][scene tables]
template <bool LDS_SCENE> RT_HD constexpr uint32_t stack_lds_bytes() { return StackCfg<LDS_SCENE>::LDS_N * 2 * RT_BLOCK * 4; }

// Dynamic LDS layout of the streaming kernels: [todo stack][prepared-ray pool: RT_POOL_BYTES][scene tables]
template <bool LDS_SCENE> RT_HD constexpr uint32_t stream_lds_bytes() { return stack_lds_bytes<LDS_SCENE>() + RT_POOL_BYTES; }

template <bool LDS_SCENE>
RT_DEV void tables_setup(const DevScene &sc, uint4 *smem, SceneTables &tb, uint32_t sceneOffsetBytes = stack_lds_bytes<LDS_SCENE>()) {
    if (LDS_SCENE) {
        stage_scene_lds(sc, smem + sceneOffsetBytes / 16, tb);
    } else {
        tb.nodes = sc.nodes;
        tb.leafIdx = sc.leafIdx;
        tb.triHot = sc.triHot;
    }
}

// ---------------------------------------------------------------------------------------------------
// batch tracing (rt_trace_closest / rt_trace_any)
// ---------------------------------------------------------------------------------------------------
// Two schedules (rt_set_option RT_OPTION_TRACE_SCHEDULE). Lockstep: one ray per lane, the wave waits for its
// slowest lane; best for coherent batches (camera rays) and small batches. Streaming (dstream.h): lanes take
// new rays as they finish; best for large batches of incoherent rays.
template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_trace_closest(DevScene sc, uint64_t n, const float *__restrict__ o, const float *__restrict__ d,
                                                            const float *__restrict__ tmax, int *__restrict__ prim, float *__restrict__ tOut,
                                                            float *__restrict__ bary, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem);
    TravCounters cn = {0, 0, 0, 0, 0};
    uint32_t myRays = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RT_BLOCK) {
        RayCtx r;
        ray_setup(r, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        HitRec h;
        kd_trace<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, tmax ? tmax[i] : RT_INF, h, cn);
        if (prim) prim[i] = h.prim;
        if (tOut) tOut[i] = h.t;
        if (bary) { bary[3 * i] = h.b0; bary[3 * i + 1] = h.b1; bary[3 * i + 2] = h.b2; }
        ++myRays;
    }
    if (STATS) flush_counters(stats, cn, myRays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)n);
}

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_trace_any(DevScene sc, uint64_t n, const float *__restrict__ o, const float *__restrict__ d,
                                                        const float *__restrict__ tmax, uint8_t *__restrict__ occluded, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem);
    TravCounters cn = {0, 0, 0, 0, 0};
    uint32_t myRays = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RT_BLOCK) {
        RayCtx r;
        ray_setup(r, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        HitRec h;
        const bool hit = kd_trace<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, tmax ? tmax[i] : RT_INF, h, cn);
        occluded[i] = hit ? 1 : 0;
        ++myRays;
    }
    if (STATS) flush_counters(stats, cn, myRays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)n);
}

// Streaming flavours. n < 2^32 (the host splits larger batches); the grid must be resident as a whole.
template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_trace_closest_stream(DevScene sc, uint64_t n, const float *__restrict__ o, const float *__restrict__ d,
                                                                   const float *__restrict__ tmax, int *__restrict__ prim, float *__restrict__ tOut,
                                                                   float *__restrict__ bary, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb, stream_lds_bytes<LDS_SCENE>());
    TravCounters cn = {0, 0, 0, 0, 0};
    TraceClosestJob job = {o, d, tmax, prim, tOut, bary};
    trace_stream<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), smem + stack_lds_bytes<LDS_SCENE>() / 16,
                                                                      (uint32_t)n, job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)n);
}

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_trace_any_stream(DevScene sc, uint64_t n, const float *__restrict__ o, const float *__restrict__ d,
                                                               const float *__restrict__ tmax, uint8_t *__restrict__ occluded, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb, stream_lds_bytes<LDS_SCENE>());
    TravCounters cn = {0, 0, 0, 0, 0};
    TraceAnyJob job = {o, d, tmax, occluded};
    trace_stream<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), smem + stack_lds_bytes<LDS_SCENE>() / 16,
                                                                       (uint32_t)n, job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)n);
}

// Refilling flavours (drefill.h): the lockstep kernels' LDS layout and grid; a wave takes its rays 64 at a time, chunk c
// by wave c mod the number of waves. n < 2^32.
#ifndef RT_RF_MIN_BLOCKS
#define RT_RF_MIN_BLOCKS 5   // workgroups per CU the register allocation of the refilling kernels is held to (8: spills; 4 and 6: slower)
#endif
template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK, RT_RF_MIN_BLOCKS) void k_trace_closest_refill(DevScene sc, uint64_t n, const float *__restrict__ o,
                                                                                     const float *__restrict__ d, const float *__restrict__ tmax,
                                                                                     int *__restrict__ prim, float *__restrict__ tOut,
                                                                                     float *__restrict__ bary, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    TravCounters cn = {0, 0, 0, 0, 0};
    TraceClosestJob job = {o, d, tmax, prim, tOut, bary};
    trace_refill<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), (uint32_t)n, job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)n);
}

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK, RT_RF_MIN_BLOCKS) void k_trace_any_refill(DevScene sc, uint64_t n, const float *__restrict__ o,
                                                                                 const float *__restrict__ d, const float *__restrict__ tmax,
                                                                                 uint8_t *__restrict__ occluded, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    TravCounters cn = {0, 0, 0, 0, 0};
    TraceAnyJob job = {o, d, tmax, occluded};
    trace_refill<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), (uint32_t)n, job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)n);
}

// ---------------------------------------------------------------------------------------------------
// wavefront path tracer
// ---------------------------------------------------------------------------------------------------
// Queue slot layouts (all float4, indexed by queue slot so a wave reads contiguous memory):
//   ray   A = {o.xyz, -}   B = {d.xyz, bits(pathId | specularBounce << 31)}   (shading a triangle hit needs B only)
//   hit     = {bits(prim), b0, b1, b2}
//   shadow A = {o.xyz, d.x}  B = {d.y, d.z, bits(pathId), c.r}  C = {c.g, c.b, -, -}
//   probe   = 6 float4: {p, hasShadow} {shadowDir, pathId} {A, light} {wi, weight} {f|cos|, pdf} {beta, selPdf}
// Per-path state (indexed by pathId = localSample * nPix + pixel):
//   L = {rgb, etaScale}   beta = {rgb, -}   rng = PCG32 state and increment
struct PathBuffers {
    float4 *rayA[2], *rayB[2];
    float4 *hit;
    float4 *L, *beta;
    uint4 *rng;   // PCG32 state (x,y) and increment (z,w)
    float4 *shA, *shB, *shC;
    float4 *probe;
    uint32_t *counts;   // see cnt_index
};

// Queue counters: one 128-byte line per counter so that appends to different queues do not serialise on one
// L2 line. Counter q of bounce b lives at counts[(b * RT_CNT_PER_BOUNCE + q) * RT_CNT_STRIDE]; q: 0 = rays entering
// bounce b, 1 = shadow entries of bounce b, 2 = probe entries of bounce b.
#define RT_CNT_STRIDE 32
#define RT_CNT_PER_BOUNCE 3
RT_HD uint32_t cnt_index(int bounce, int q) { return (uint32_t)(bounce * RT_CNT_PER_BOUNCE + q) * RT_CNT_STRIDE; }

struct PassInfo {
    uint32_t nPix;         // pixels in the sample bounds
    uint32_t sbW;          // sample-bounds width
    uint32_t sppPass;      // samples per pixel in this pass
    uint32_t sppBase;      // global index of local sample 0
    uint32_t sppTotal;
    uint64_t seed;
    int maxDepth;
    float rrThreshold;
};

RT_DEV uint64_t path_sequence(const CameraDev &cam, const PassInfo &ps, uint32_t pathId, int &px, int &py, uint32_t &sLocal) {
    sLocal = pathId / ps.nPix;
    const uint32_t pi = pathId - sLocal * ps.nPix;
    const uint32_t ly = pi / ps.sbW, lx = pi - ly * ps.sbW;
    px = cam.sb[0] + (int)lx;
    py = cam.sb[1] + (int)ly;
    const uint64_t pix = (uint64_t)py * (uint64_t)cam.width + (uint64_t)px;
    return counter_sequence(pix, ps.sppTotal, ps.sppBase + sLocal, ps.seed);
}

// Block-aggregated append to two queues at once: wave ballots give per-wave counts, one lane per queue issues a
// single atomicAdd for the whole 256-thread block, and every lane gets base + rank. All threads of the block must
// call it (it synchronises). `lds` holds 16 dwords.
RT_DEV void block_append2(uint32_t *lds, uint32_t *c0, uint32_t *c1, bool w0, bool w1, uint32_t &s0, uint32_t &s1) {
    static_assert(RT_BLOCK == 256, "the per-wave counts below are summed over exactly four waves");
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long m0 = __ballot(w0), m1 = __ballot(w1);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
        lds[0 * 4 + wave] = (uint32_t)__popcll(m0);
        lds[1 * 4 + wave] = (uint32_t)__popcll(m1);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint32_t q = threadIdx.x;
        const uint32_t total = lds[q * 4 + 0] + lds[q * 4 + 1] + lds[q * 4 + 2] + lds[q * 4 + 3];
        uint32_t *ctr = q == 0 ? c0 : c1;
        lds[12 + q] = total ? atomicAdd(ctr, total) : 0u;
    }
    __syncthreads();
    uint32_t pre0 = 0, pre1 = 0;
    for (uint32_t w = 0; w < wave; ++w) { pre0 += lds[0 * 4 + w]; pre1 += lds[1 * 4 + w]; }
    s0 = lds[12] + pre0 + (uint32_t)__popcll(m0 & below);
    s1 = lds[13] + pre1 + (uint32_t)__popcll(m1 & below);
    __syncthreads();   // lds is reused by the next iteration
}

__global__ __launch_bounds__(RT_BLOCK) void k_raygen(CameraDev cam, PassInfo ps, PathBuffers pb, unsigned long long *stats) {
    const uint32_t n = ps.nPix * ps.sppPass;
    for (uint32_t id = blockIdx.x * RT_BLOCK + threadIdx.x; id < n; id += gridDim.x * RT_BLOCK) {
        int px, py;
        uint32_t sLocal;
        Pcg rng;
        pcg_set_sequence(rng, path_sequence(cam, ps, id, px, py, sLocal));
        const float u0 = pcg_float(rng), u1 = pcg_float(rng);
        f3 o, d;
        camera_ray(cam, (float)px + u0, (float)py + u1, o, d);
        // every camera ray starts at the pinhole: no origin records at bounce 0 (camera_origin)
        pb.rayB[0][id] = make_float4(d.x, d.y, d.z, __uint_as_float(id));
        pb.L[id] = make_float4(0.f, 0.f, 0.f, 1.f);
        // beta = 1 and the generator's state after the two film draws are rebuilt by k_shade at bounce 0 (half of the
        // camera rays of a 16:9 cornellBox frame miss and never need them)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        pb.counts[cnt_index(0, 0)] = n;
        atomicAdd(&stats[ST_CAMERA_SAMPLES], (unsigned long long)n);
    }
}

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_extend(DevScene sc, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB,
                                                     float4 *__restrict__ hit, const uint32_t *__restrict__ count, unsigned long long *stats,
                                                     float4 commonOrigin) {
    // rayA == nullptr: all rays start at commonOrigin (camera rays)
    RT_DYNAMIC_LDS(smem);
    const uint32_t n = *count;
    if (blockIdx.x * RT_BLOCK >= n) return;   // whole block idle: skip the LDS staging
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem);
    TravCounters cn = {0, 0, 0, 0, 0};
    uint32_t myRays = 0;
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) {
        const float4 a = rayA ? rayA[i] : commonOrigin, b = rayB[i];
        RayCtx r;
        ray_setup(r, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z));
        HitRec h;
        kd_trace<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, RT_INF, h, cn);
        hit[i] = make_float4(__int_as_float(h.prim), h.b0, h.b1, h.b2);
        ++myRays;
    }
    if (STATS) flush_counters(stats, cn, myRays);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)n);
        atomicAdd(&stats[ST_EXTEND_RAYS], (unsigned long long)n);
    }
}

// Streaming flavour of k_extend, used for scenes whose tables live in HBM: there a ray costs a hundred dependent
// node fetches and the gap between the slowest and the average lane of a wave is what bounds the kernel.
struct ExtendJob {
    const float4 *rayA, *rayB;
    float4 *hit;
    float4 commonOrigin;   // used when rayA is null
    RT_DEV bool fetch(uint32_t j, f3 &ro, f3 &rd, float &tm) const {
        const float4 a = rayA ? rayA[j] : commonOrigin, b = rayB[j];
        ro = mk3(a.x, a.y, a.z);
        rd = mk3(b.x, b.y, b.z);
        tm = RT_INF;
        return true;
    }
    RT_DEV void finish(uint32_t j, const HitRec &h) const { hit[j] = make_float4(__int_as_float(h.prim), h.b0, h.b1, h.b2); }
};

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_extend_stream(DevScene sc, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB,
                                                            float4 *__restrict__ hit, const uint32_t *__restrict__ count,
                                                            unsigned long long *stats, float4 commonOrigin) {
    RT_DYNAMIC_LDS(smem);
    const uint32_t n = *count;
    if (blockIdx.x * RT_BLOCK >= n) return;
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb, stream_lds_bytes<LDS_SCENE>());
    TravCounters cn = {0, 0, 0, 0, 0};
    ExtendJob job = {rayA, rayB, hit, commonOrigin};
    trace_stream<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), smem + stack_lds_bytes<LDS_SCENE>() / 16, n,
                                                                      job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)n);
        atomicAdd(&stats[ST_EXTEND_RAYS], (unsigned long long)n);
    }
}

// Refilling flavour of k_extend for scenes in LDS (drefill.h): same LDS layout and grid as k_extend.
template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK, RT_RF_MIN_BLOCKS) void k_extend_refill(DevScene sc, const float4 *__restrict__ rayA,
                                                                              const float4 *__restrict__ rayB, float4 *__restrict__ hit,
                                                                              const uint32_t *__restrict__ count, unsigned long long *stats,
                                                                              float4 commonOrigin) {
    RT_DYNAMIC_LDS(smem);
    const uint32_t n = *count;
    if (blockIdx.x * RT_BLOCK >= n) return;
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    TravCounters cn = {0, 0, 0, 0, 0};
    ExtendJob job = {rayA, rayB, hit, commonOrigin};
    trace_refill<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), n, job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)n);
        atomicAdd(&stats[ST_EXTEND_RAYS], (unsigned long long)n);
    }
}

// One bounce of PathRenderer::Li (Render/Render.cpp:363-445) for every live path, after its closest hit
// is known: emission, light sampling with MIS (uniformSampleOneLight / estimateDirect, :189-312), BSDF
// sampling, Russian roulette. Visibility is resolved by k_shadow / k_probe afterwards.
#ifndef RT_SHADE_MIN_BLOCKS
#define RT_SHADE_MIN_BLOCKS 2   // workgroups per CU the register allocation of k_shade is held to
#endif
// The tables shading reads (primitive records, shading records, materials, lights, light selection tables, spheres) as one
// block of 16-byte vectors in LDS, for scenes where that is a few kilobytes: a shaded hit follows a chain of dependent reads
// through them (primitive -> material, light choice -> light -> its primitive), which costs a memory round trip per link
// from HBM or L2 and a few dozen cycles from LDS.
RT_HD uint32_t shade_tables_vecs(uint32_t nPrims, uint32_t nMats, uint32_t nLights, uint32_t nSpheres) {
    return nPrims * 3u + nPrims * 5u + nMats * 2u + nLights * 2u + (nLights + 1u + 3u) / 4u + (nLights + 3u) / 4u + nSpheres * 9u;
}
RT_DEV void stage_shade_tables(DevScene &sc, uint4 *smem) {
    const uint32_t counts[7] = {sc.nPrims * 3u, sc.nPrims * 5u, sc.nMats * 2u, sc.nLights * 2u, (sc.nLights + 1u + 3u) / 4u,
                                (sc.nLights + 3u) / 4u, sc.nSpheres * 9u};
    const uint4 *src[7] = {reinterpret_cast<const uint4 *>(sc.triHot),   reinterpret_cast<const uint4 *>(sc.triShade),
                           reinterpret_cast<const uint4 *>(sc.mats),     reinterpret_cast<const uint4 *>(sc.lights),
                           reinterpret_cast<const uint4 *>(sc.lightCdf), reinterpret_cast<const uint4 *>(sc.lightFunc),
                           reinterpret_cast<const uint4 *>(sc.spheres)};
    uint32_t at[8];
    at[0] = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        for (uint32_t i = threadIdx.x; i < counts[k]; i += blockDim.x) smem[at[k] + i] = src[k][i];
        at[k + 1] = at[k] + counts[k];
    }
    sc.triHot = reinterpret_cast<const float4 *>(smem + at[0]);
    sc.triShade = reinterpret_cast<const float4 *>(smem + at[1]);
    sc.mats = reinterpret_cast<const DevMaterial *>(smem + at[2]);
    sc.lights = reinterpret_cast<const DevLight *>(smem + at[3]);
    sc.lightCdf = reinterpret_cast<const float *>(smem + at[4]);
    sc.lightFunc = reinterpret_cast<const float *>(smem + at[5]);
    sc.spheres = reinterpret_cast<const DevSphere *>(smem + at[6]);
    __syncthreads();
}

// LDS_TABLES: the launch gives 16 * shade_tables_vecs(...) bytes of dynamic LDS and the kernel shades out of a copy there.
template <int MATSET, bool SPH, bool LDS_TABLES>
__global__ __launch_bounds__(RT_BLOCK, RT_SHADE_MIN_BLOCKS) void k_shade(DevScene scene, CameraDev cam, PassInfo ps, PathBuffers pb, int bounce) {
    const int cur = bounce & 1, nxt = cur ^ 1;
    __shared__ uint32_t s_app[16];
    const uint32_t n = pb.counts[cnt_index(bounce, 0)];
    if (blockIdx.x * RT_BLOCK >= n) return;   // no chunk for this workgroup
    DevScene sc = scene;
    if (LDS_TABLES) {
        RT_DYNAMIC_LDS(smem);
        stage_shade_tables(sc, smem);
    }
    uint32_t *cntNext = &pb.counts[cnt_index(bounce + 1, 0)];
    uint32_t *cntShadow = &pb.counts[cnt_index(bounce, 1)];
    uint32_t *cntProbe = &pb.counts[cnt_index(bounce, 2)];
    // Rays that missed have nothing to shade. The block reads hit records and direction records 256 at a time, keeps those of
    // the hits in a ring in LDS and shades 256 of them at a time, so that (but for the block's last round) every lane has
    // work. The records of the next chunk are requested before a round's shading and put into the ring after it, so their
    // memory round trip runs beside the shading instead of in front of it.
    __shared__ uint32_t s_fill[8];
    __shared__ uint32_t s_slot[2 * RT_BLOCK];
    __shared__ float4 s_hit[2 * RT_BLOCK];
    __shared__ float4 s_rb[2 * RT_BLOCK];
    const uint32_t nChunks = (n + RT_BLOCK - 1u) / RT_BLOCK;
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    uint32_t chunk = blockIdx.x, head = 0, pending = 0, parity = 0;   // the same in every thread of the block
    // the chunk on its way: requested, not yet in the ring
    float4 hjP = make_float4(0.f, 0.f, 0.f, 0.f), rbP = hjP;
    uint32_t jP = 0;
    bool onTheWay = false;   // the same in every thread of the block
    auto request = [&]() {
        jP = chunk * RT_BLOCK + threadIdx.x;
        chunk += gridDim.x;
        hjP = rbP = make_float4(-0.f, 0.f, 0.f, 0.f);   // the sign bit of x: a record that is no hit
        if (jP < n) {
            hjP = pb.hit[jP];
            rbP = pb.rayB[cur][jP];
        }
        onTheWay = true;
    };
    if (chunk < nChunks) request();
    for (;;) {
        while (pending < RT_BLOCK && (onTheWay || chunk < nChunks)) {
            if (!onTheWay) request();
            const float4 hj = hjP, rbj = rbP;
            const uint32_t j = jP;
            onTheWay = false;
            const bool isHit = __float_as_int(hj.x) >= 0;
            const unsigned long long m = __ballot(isHit);
            if (lane == 0) s_fill[parity * 4 + wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t pre = 0, total = 0;
            for (uint32_t w = 0; w < RT_BLOCK / 64; ++w) {
                const uint32_t cw = s_fill[parity * 4 + w];
                pre += w < wave ? cw : 0u;
                total += cw;
            }
            if (isHit) {
                const uint32_t e = (head + pending + pre + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (2 * RT_BLOCK - 1u);
                s_slot[e] = j;
                s_hit[e] = hj;
                s_rb[e] = rbj;
            }
            pending += total;
            parity ^= 1u;   // the next round's counts go to the other half: one barrier per round is enough
        }
        if (pending == 0) break;
        if (chunk < nChunks) request();   // consumed after this round's shading
        const uint32_t take = pending < RT_BLOCK ? pending : RT_BLOCK;
        __syncthreads();   // ring entries written above are visible
        bool alive = false;   // set when the path continues
        bool wantShadow = false;
        float4 sA, sB, sC;          // shadow entry
        float4 nA, nB;              // next ray
        sA = sB = sC = nA = nB = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x < take) {
            const uint32_t e = (head + threadIdx.x) & (2 * RT_BLOCK - 1u);
            const uint32_t i = s_slot[e];
            const float4 hr = s_hit[e];
            const float4 rb = s_rb[e];
            // only a sphere hit needs the origin
            const float4 ra = SPH ? (bounce > 0 ? pb.rayA[cur][i] : camera_origin(cam)) : make_float4(0.f, 0.f, 0.f, 0.f);
            const uint32_t idBits = __float_as_uint(rb.w);
            const uint32_t pathId = idBits & 0x7fffffffu;
            const bool specularBounce = (idBits >> 31) != 0;
            const f3 rd = mk3(rb.x, rb.y, rb.z);
            const int prim = __float_as_int(hr.x);
            {
                Surf sf;
                surface_at<SPH>(sc, prim, hr.y, hr.z, hr.w, mk3(ra.x, ra.y, ra.z), rd, sf);
                // L is read for an emission to add and for etaScale; a scene of Lambertian materials has neither specular
                // bounces nor refraction (etaScale stays 1), so there only a camera ray that sees a light needs it
                const bool emits = (bounce == 0 || specularBounce) && sf.light >= 0;
                float4 L4 = make_float4(0.f, 0.f, 0.f, 1.f);
                if (MATSET != MATSET_LAMBERT || emits) L4 = pb.L[pathId];
                f3 beta = mk3(1.f, 1.f, 1.f);
                if (bounce > 0) {
                    const float4 beta4 = pb.beta[pathId];
                    beta = mk3(beta4.x, beta4.y, beta4.z);
                }
                float etaScale = L4.w;
                bool Ldirty = false;
                if (emits) {
                    const f3 le = light_L(sc.lights[sf.light], sf.n, -rd);
                    const f3 add = beta * le;
                    L4.x += add.x; L4.y += add.y; L4.z += add.z;
                    Ldirty = true;
                }
                if (bounce < ps.maxDepth) {
                    DBsdf bs;
                    bsdf_init<MATSET>(bs, sf, sc.mats[sf.mat]);
                    Pcg rng;
                    uint4 st;
                    if (bounce > 0) {
                        st = pb.rng[pathId];
                        rng.state = (uint64_t)st.x | ((uint64_t)st.y << 32);
                        rng.inc = (uint64_t)st.z | ((uint64_t)st.w << 32);
                    } else {   // the stream k_raygen drew the film position from
                        int px, py;
                        uint32_t sLocal;
                        pcg_set_sequence(rng, path_sequence(cam, ps, pathId, px, py, sLocal));
                        (void)pcg_float(rng);
                        (void)pcg_float(rng);
                        st.z = (uint32_t)rng.inc;
                        st.w = (uint32_t)(rng.inc >> 32);
                    }
                    const f3 isectWo = normalize3(-rd);

                    if (bsdf_num(bs, BSDF_ALL & ~BSDF_SPECULAR) > 0 && sc.nLights > 0) {
                        // uniformSampleOneLight
                        float selPdf;
                        const int li = sample_light(sc, pcg_float(rng), selPdf);
                        if (selPdf != 0) {
                            const float ul0 = pcg_float(rng), ul1 = pcg_float(rng);
                            const float us0 = pcg_float(rng), us1 = pcg_float(rng);
                            const DevLight lt = sc.lights[li];
                            const int flags = BSDF_ALL & ~BSDF_SPECULAR;
                            // estimateDirect, light-sampling half
                            f3 wi, pShape;
                            float lightPdf = 0, scatteringPdf = 0;
                            f3 Li = light_sample_Li<SPH>(sc, lt, sf.p, ul0, ul1, wi, lightPdf, pShape);
                            f3 A = mk3(0.f, 0.f, 0.f);
                            bool haveShadow = false;
                            f3 shD = mk3(0.f, 0.f, 0.f);
                            if (lightPdf > 0 && !black3(Li)) {
                                const f3 f = bsdf_f(bs, isectWo, wi, flags) * absdot3(wi, sf.n);
                                scatteringPdf = bsdf_pdf(bs, isectWo, wi, flags);
                                if (!black3(f)) {
                                    const float weight = power_heuristic(lightPdf, scatteringPdf);
                                    A = f * Li * weight / lightPdf;
                                    haveShadow = true;
                                    shD = normalize3(pShape - sf.p);   // Interaction::spawnRayTo -> Ray constructor
                                }
                            }
                            // estimateDirect, BSDF-sampling half
                            bool needProbe = false;
                            f3 wi2 = mk3(0.f, 0.f, 0.f), fB = mk3(0.f, 0.f, 0.f);
                            float weightB = 1, pdfB = scatteringPdf;
                            {
                                int sampledType = 0;
                                f3 f2 = bsdf_sample(bs, isectWo, wi2, us0, us1, pdfB, sampledType, flags);
                                f2 = f2 * absdot3(wi2, sf.n);
                                const bool sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
                                if (!black3(f2) && pdfB > 0) {
                                    bool go = true;
                                    if (!sampledSpecular) {
                                        const float lp = light_pdf_Li<SPH>(sc, lt, sf.p, wi2);
                                        if (lp == 0) go = false;
                                        else weightB = power_heuristic(pdfB, lp);
                                    }
                                    if (go) { needProbe = true; fB = f2; }
                                }
                            }
                            if (needProbe) {
                                // rare (the BSDF sample has to point at the chosen light): appended right here with one
                                // atomic per lane instead of being carried in 24 registers to the block-wide append
                                float4 *pr = pb.probe + (size_t)atomicAdd(cntProbe, 1u) * 6;
                                pr[0] = make_float4(sf.p.x, sf.p.y, sf.p.z, haveShadow ? 1.f : 0.f);
                                pr[1] = make_float4(shD.x, shD.y, shD.z, __uint_as_float(pathId));
                                pr[2] = make_float4(A.x, A.y, A.z, __int_as_float(li));
                                pr[3] = make_float4(wi2.x, wi2.y, wi2.z, weightB);
                                pr[4] = make_float4(fB.x, fB.y, fB.z, pdfB);
                                pr[5] = make_float4(beta.x, beta.y, beta.z, selPdf);
                            } else if (haveShadow) {
                                // Ld = (0 + A) / selPdf, L += beta * Ld once the light is known to be visible
                                const f3 c = beta * (A / selPdf);
                                wantShadow = true;
                                sA = make_float4(sf.p.x, sf.p.y, sf.p.z, shD.x);
                                sB = make_float4(shD.y, shD.z, __uint_as_float(pathId), c.x);
                                sC = make_float4(c.y, c.z, 0.f, 0.f);
                            }
                        }
                    }

                    // sample the BSDF for the next direction
                    {
                        const f3 wo = -rd;
                        f3 wi = mk3(0.f, 0.f, 0.f);
                        float pdf = 0;
                        int flags = 0;
                        const float u0 = pcg_float(rng), u1 = pcg_float(rng);
                        const f3 f = bsdf_sample(bs, wo, wi, u0, u1, pdf, flags, BSDF_ALL);
                        if (!(black3(f) || pdf == 0.f)) {
                            beta = beta * (f * absdot3(wi, sf.n) / pdf);
                            const bool spec = (flags & BSDF_SPECULAR) != 0;
                            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                                const float eta = bs.eta;
                                etaScale *= (dot3(wo, sf.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                            }
                            const f3 nd = normalize3(wi);   // Interaction::spawnRay -> Ray constructor
                            alive = true;
                            const f3 rrBeta = beta * etaScale;
                            const float mx = gmax(gmax(rrBeta.x, rrBeta.y), rrBeta.z);
                            if (mx < ps.rrThreshold && bounce > 3) {
                                const float q = gmax(.05f, 1 - mx);
                                if (pcg_float(rng) < q) alive = false;
                                else beta = beta / (1 - q);
                            }
                            if (alive) {
                                nA = make_float4(sf.p.x, sf.p.y, sf.p.z, 0.f);
                                nB = make_float4(nd.x, nd.y, nd.z, __uint_as_float(pathId | (spec ? 0x80000000u : 0u)));
                                pb.beta[pathId] = make_float4(beta.x, beta.y, beta.z, 0.f);
                                pb.rng[pathId] = make_uint4((uint32_t)rng.state, (uint32_t)(rng.state >> 32), st.z, st.w);
                                if (etaScale != L4.w) { L4.w = etaScale; Ldirty = true; }
                            }
                        }
                    }
                }
                if (Ldirty) pb.L[pathId] = L4;
            }
        }
        head += take;
        pending -= take;
        uint32_t slotN, slotS;
        block_append2(s_app, cntNext, cntShadow, alive, wantShadow, slotN, slotS);
        if (alive) { pb.rayA[nxt][slotN] = nA; pb.rayB[nxt][slotN] = nB; }
        if (wantShadow) { pb.shA[slotS] = sA; pb.shB[slotS] = sB; pb.shC[slotS] = sC; }
    }
}

// Shadow rays of the light-sampling half (VisibilityTester::unoccluded, Render/Light.cpp:28-31): any-hit
// over Ray(p, pLight - p, 1 - ShadowEpsilon); a visible light adds its contribution to the path's L.
template <bool LDS_SCENE, bool STATS, bool SPH>
RT_DEV void shadow_entries(const DevScene &sc, const SceneTables &tb, uint32_t *stk, const PathBuffers &pb, uint32_t n, uint32_t block,
                           uint32_t nBlocks, unsigned long long *stats) {
    TravCounters cn = {0, 0, 0, 0};
    uint32_t myRays = 0;
    for (uint32_t i = block * RT_BLOCK + threadIdx.x; i < n; i += nBlocks * RT_BLOCK) {
        const float4 a = pb.shA[i], b = pb.shB[i], c = pb.shC[i];
        RayCtx r;
        ray_setup(r, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
        HitRec h;
        const bool occluded = kd_trace<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, 1 - RT_SHADOW_EPS, h, cn);
        if (!occluded) {
            const uint32_t pathId = __float_as_uint(b.z);
            float4 L4 = pb.L[pathId];
            L4.x += b.w; L4.y += c.x; L4.z += c.y;
            pb.L[pathId] = L4;
        }
        ++myRays;
    }
    if (STATS) flush_counters(stats, cn, myRays);
    if (block == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)n);
}

struct ShadowJob {
    const float4 *shA, *shB, *shC;
    float4 *L;
    RT_DEV bool fetch(uint32_t j, f3 &ro, f3 &rd, float &tm) const {
        const float4 a = shA[j], b = shB[j];
        ro = mk3(a.x, a.y, a.z);
        rd = mk3(a.w, b.x, b.y);
        tm = 1 - RT_SHADOW_EPS;
        return true;
    }
    RT_DEV void finish(uint32_t j, const HitRec &h) const {
        if (h.prim >= 0) return;   // occluded
        const float4 b = shB[j], c = shC[j];
        const uint32_t pathId = __float_as_uint(b.z);
        float4 L4 = L[pathId];
        L4.x += b.w; L4.y += c.x; L4.z += c.y;
        L[pathId] = L4;
    }
};

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_shadow_stream(DevScene sc, PathBuffers pb, int bounce, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    const uint32_t n = pb.counts[cnt_index(bounce, 1)];
    if (blockIdx.x * RT_BLOCK >= n) return;
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb, stream_lds_bytes<LDS_SCENE>());
    TravCounters cn = {0, 0, 0, 0, 0};
    ShadowJob job = {pb.shA, pb.shB, pb.shC, pb.L};
    trace_stream<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), smem + stack_lds_bytes<LDS_SCENE>() / 16, n,
                                                                       job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)n);
}

// Refilling flavour of k_shadow (drefill.h), for scenes in HBM; on scenes in LDS the lockstep k_shadow is faster (the rays
// are 0.9999 long and their cost is the set-up, which lockstep runs at full width).
template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK, RT_RF_MIN_BLOCKS) void k_shadow_refill(DevScene sc, PathBuffers pb, int bounce, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    const uint32_t n = pb.counts[cnt_index(bounce, 1)];
    if (blockIdx.x * RT_BLOCK >= n) return;
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    TravCounters cn = {0, 0, 0, 0, 0};
    ShadowJob job = {pb.shA, pb.shB, pb.shC, pb.L};
    trace_refill<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), n, job, cn);
    if (STATS) flush_counters(stats, cn, cn.rays);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)n);
}

// Paths whose BSDF sample may see the chosen light (estimateDirect, Render/Render.cpp:266-310): resolve the
// shadow ray of the light sample (if any) and the closest hit along the BSDF sample, then add
// beta * ((A' + B') / selPdf) to the path's L with the reference's association.
template <bool LDS_SCENE, bool STATS, bool SPH>
RT_DEV void probe_entries(const DevScene &sc, const SceneTables &tb, uint32_t *stk, const PathBuffers &pb, uint32_t n, uint32_t block,
                          uint32_t nBlocks, unsigned long long *stats) {
    TravCounters cn = {0, 0, 0, 0};
    uint32_t myRays = 0, nShadow = 0;
    for (uint32_t i = block * RT_BLOCK + threadIdx.x; i < n; i += nBlocks * RT_BLOCK) {
        const float4 e0 = pb.probe[(size_t)i * 6 + 0], e1 = pb.probe[(size_t)i * 6 + 1], e2 = pb.probe[(size_t)i * 6 + 2];
        const float4 e3 = pb.probe[(size_t)i * 6 + 3], e4 = pb.probe[(size_t)i * 6 + 4], e5 = pb.probe[(size_t)i * 6 + 5];
        const f3 p = mk3(e0.x, e0.y, e0.z);
        const uint32_t pathId = __float_as_uint(e1.w);
        const int li = __float_as_int(e2.w);
        f3 Ld = mk3(0.f, 0.f, 0.f);
        if (e0.w != 0.f) {
            RayCtx r;
            ray_setup(r, p, mk3(e1.x, e1.y, e1.z));
            HitRec h;
            const bool occluded = kd_trace<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, 1 - RT_SHADOW_EPS, h, cn);
            if (!occluded) Ld = Ld + mk3(e2.x, e2.y, e2.z);
            ++myRays;
            ++nShadow;
        }
        {
            const f3 wi = mk3(e3.x, e3.y, e3.z);
            RayCtx r;
            ray_setup(r, p, normalize3(wi));
            HitRec h;
            const bool found = kd_trace<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, RT_INF, h, cn);
            ++myRays;
            f3 Li = mk3(0.f, 0.f, 0.f);
            if (found) {
                Surf sf;
                surface_at<SPH>(sc, h.prim, h.b0, h.b1, h.b2, p, r.d, sf);
                if (sf.light == li) Li = light_L(sc.lights[li], sf.n, -wi);
            }
            if (!black3(Li)) Ld = Ld + mk3(e4.x, e4.y, e4.z) * Li * mk3(1.f, 1.f, 1.f) * e3.w / e4.w;
        }
        const f3 add = mk3(e5.x, e5.y, e5.z) * (Ld / e5.w);
        float4 L4 = pb.L[pathId];
        L4.x += add.x; L4.y += add.y; L4.z += add.z;
        pb.L[pathId] = L4;
    }
    if (STATS) flush_counters(stats, cn, myRays);
    if (nShadow) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)nShadow);
    if (block == 0 && threadIdx.x == 0) atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)n);
}

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_shadow(DevScene sc, PathBuffers pb, int bounce, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    const uint32_t n = pb.counts[cnt_index(bounce, 1)];
    if (blockIdx.x * RT_BLOCK >= n) return;
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    shadow_entries<LDS_SCENE, STATS, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), pb, n, blockIdx.x, gridDim.x, stats);
}

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_probe(DevScene sc, PathBuffers pb, int bounce, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    const uint32_t n = pb.counts[cnt_index(bounce, 2)];
    if (blockIdx.x * RT_BLOCK >= n) return;
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    probe_entries<LDS_SCENE, STATS, SPH>(sc, tb, reinterpret_cast<uint32_t *>(smem), pb, n, blockIdx.x, gridDim.x, stats);
}

// Sample filter of the render loop (Render/Render.cpp:72-83).
RT_DEV f3 sanitize_L(f3 L) {
    const float lum = 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z;
    if (isnan(L.x) || isnan(L.y) || isnan(L.z)) return mk3(0.f, 0.f, 0.f);
    if ((double)lum < -1e-5) return mk3(0.f, 0.f, 0.f);
    if (isinf(lum)) return mk3(0.f, 0.f, 0.f);
    return L;
}

// FilmTile::addSample for the Box filter (Object/Film.h:111-155; the filter table holds 1 everywhere, Render/Filter.cpp:20-23):
// pixels [x0,x1) x [y0,y1), clipped to the cropped pixel bounds, receive L with weight 1. Returns true when that set is
// exactly the sample's own pixel (always, for radius 0.5, unless a jitter is exactly 0).
RT_DEV bool film_footprint(const CameraDev &cam, float pfx, float pfy, int px, int py, int &x0, int &y0, int &x1, int &y1) {
    const float dx = pfx - 0.5f, dy = pfy - 0.5f;
    x0 = max((int)ceilf(dx - cam.filterRadius[0]), cam.cb[0]);
    y0 = max((int)ceilf(dy - cam.filterRadius[1]), cam.cb[1]);
    x1 = min((int)floorf(dx + cam.filterRadius[0]) + 1, cam.cb[2]);
    y1 = min((int)floorf(dy + cam.filterRadius[1]) + 1, cam.cb[3]);
    return x0 == px && y0 == py && x1 == px + 1 && y1 == py + 1;
}
// "MaxLum" of the film (Object/Film.h:113-115)
RT_DEV f3 film_clamp(const CameraDev &cam, f3 L) {
    const float lum = 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z;
    if (lum > cam.maxSampleLuminance) L = L * (cam.maxSampleLuminance / lum);
    return L;
}
RT_DEV void film_spill(const CameraDev &cam, float *spillRgb, float *spillW, int px, int py, int x0, int y0, int x1, int y1, f3 L,
                       f3 &own, float &ownW) {
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            if (x == px && y == py) { own = own + L; ownW += 1.0f; continue; }
            const size_t g = (size_t)y * cam.width + x;
            atomicAdd(&spillRgb[3 * g + 0], L.x);
            atomicAdd(&spillRgb[3 * g + 1], L.y);
            atomicAdd(&spillRgb[3 * g + 2], L.z);
            atomicAdd(&spillW[g], 1.0f);
        }
}

// End of a pass: every pixel adds its samples of this pass to the film in sample order.
__global__ __launch_bounds__(RT_BLOCK) void k_film_path(CameraDev cam, PassInfo ps, PathBuffers pb, float *rgbSum, float *weightSum,
                                                        float *spillRgb, float *spillW) {
    for (uint32_t pi = blockIdx.x * RT_BLOCK + threadIdx.x; pi < ps.nPix; pi += gridDim.x * RT_BLOCK) {
        const uint32_t ly = pi / ps.sbW, lx = pi - ly * ps.sbW;
        const int px = cam.sb[0] + (int)lx, py = cam.sb[1] + (int)ly;
        // with a filter radius above 0.5 the sample bounds reach outside the cropped pixel bounds: such a pixel has
        // samples but no film entry of its own
        const bool onFilm = px >= cam.cb[0] && px < cam.cb[2] && py >= cam.cb[1] && py < cam.cb[3];
        const size_t g = onFilm ? (size_t)py * cam.width + px : 0;
        f3 acc = mk3(0.f, 0.f, 0.f);
        float w = 0.f;
        if (onFilm) {
            acc = mk3(rgbSum[3 * g], rgbSum[3 * g + 1], rgbSum[3 * g + 2]);
            w = weightSum[g];
        }
        for (uint32_t s = 0; s < ps.sppPass; ++s) {
            const uint32_t id = s * ps.nPix + pi;
            const float4 L4 = pb.L[id];
            const f3 L = film_clamp(cam, sanitize_L(mk3(L4.x, L4.y, L4.z)));
            const uint64_t pix = (uint64_t)py * (uint64_t)cam.width + (uint64_t)px;
            Pcg rng;
            pcg_set_sequence(rng, counter_sequence(pix, ps.sppTotal, ps.sppBase + s, ps.seed));
            const float u0 = pcg_float(rng), u1 = pcg_float(rng);
            int x0, y0, x1, y1;
            if (film_footprint(cam, (float)px + u0, (float)py + u1, px, py, x0, y0, x1, y1)) {
                acc = acc + L;
                w += 1.0f;
            } else {
                film_spill(cam, spillRgb, spillW, px, py, x0, y0, x1, y1, L, acc, w);
            }
        }
        if (onFilm) {
            rgbSum[3 * g] = acc.x; rgbSum[3 * g + 1] = acc.y; rgbSum[3 * g + 2] = acc.z;
            weightSum[g] = w;
        }
    }
}

__global__ __launch_bounds__(RT_BLOCK) void k_merge_spill(size_t nPix, float *rgbSum, float *weightSum, float *spillRgb, float *spillW) {
    for (size_t g = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; g < nPix; g += (size_t)gridDim.x * RT_BLOCK) {
        const float w = spillW[g];
        if (w != 0.f) {
            rgbSum[3 * g] += spillRgb[3 * g]; rgbSum[3 * g + 1] += spillRgb[3 * g + 1]; rgbSum[3 * g + 2] += spillRgb[3 * g + 2];
            weightSum[g] += w;
            spillRgb[3 * g] = spillRgb[3 * g + 1] = spillRgb[3 * g + 2] = 0.f;
            spillW[g] = 0.f;
        }
    }
}

// Film development: the arithmetic of Film::mergeFilmTile + Film::writeImageToFile for splat-free films
// (Object/Film.cpp:98-167): RGB -> XYZ at merge, XYZ -> RGB, divide by the filter weight sum (clamped at 0), scale,
// sRGB curve (Utils/Base.h:50-55), quantise to 8 bits. The reference's powf becomes a double pow rounded once.
__global__ __launch_bounds__(RT_BLOCK) void k_develop(size_t nPix, const float *__restrict__ rgbSum, const float *__restrict__ weightSum,
                                                      float filmScale, uint8_t *__restrict__ rgb8) {
    for (size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < nPix; i += (size_t)gridDim.x * RT_BLOCK) {
        const float c0 = rgbSum[3 * i], c1 = rgbSum[3 * i + 1], c2 = rgbSum[3 * i + 2];
        const float x = 0.412453f * c0 + 0.357580f * c1 + 0.180423f * c2;
        const float y = 0.212671f * c0 + 0.715160f * c1 + 0.072169f * c2;
        const float z = 0.019334f * c0 + 0.119193f * c1 + 0.950227f * c2;
        float rgb[3];
        rgb[0] = 3.240479f * x - 1.537150f * y - 0.498535f * z;
        rgb[1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
        rgb[2] = 0.055648f * x - 0.204043f * y + 1.057311f * z;
        const float w = weightSum[i];
        if (w != 0) {
            const float invWt = 1.0f / w;
#pragma unroll
            for (int k = 0; k < 3; ++k) rgb[k] = gmax(0.0f, rgb[k] * invWt);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = rgb[k] + 1.0f * 0.0f;
            v *= filmScale;
            const float g = (v <= 0.0031308f) ? 12.92f * v : 1.055f * (float)pow((double)v, (double)(float)(1.f / 2.4f)) - 0.055f;
            float b = 255.f * g + 0.5f;
            b = b < 0.f ? 0.f : (b > 255.f ? 255.f : b);
            rgb8[3 * i + k] = (uint8_t)b;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Whitted integrator (Render/Render.cpp:106-152, 473-532)
// ---------------------------------------------------------------------------------------------------
// Frames of the unrolled recursion. rt_render refuses a Whitted job with max_depth above this (RT_ERR_UNSUPPORTED), so
// the `sp < RT_WHITTED_MAX_FRAMES` test in whitted_li never decides a result; it only keeps fr[] in bounds.
#define RT_WHITTED_MAX_FRAMES 32

struct WFrame {
    f3 L, f;
    float absd, pdf;
    int prim;
    float b0, b1, b2;
    f3 ro, rd;
    int stage;
};

// WhittedRenderer::Li with the recursion unrolled into explicit frames; evaluates children depth-first in
// the reference's order (reflection before transmission) and combines them with its association
// (L += f * Li(child) * |wi.n| / pdf). The frame at stack position k is the call at recursion depth k.
// When `countOnly` is set nothing is shaded: the function only consumes the random numbers the full
// evaluation would consume (the specular tree does not depend on them).
template <bool LDS_SCENE, bool STATS, bool SPH>
RT_DEV f3 whitted_li(const DevScene &sc, const SceneTables &tb, uint32_t *stk, f3 ro, f3 rdir, Pcg &rng, int maxDepth, bool countOnly,
                     TravCounters &cn, uint32_t &nClosest, uint32_t &nAny, uint32_t &nDraws) {
    WFrame fr[RT_WHITTED_MAX_FRAMES];
    int sp = 0;
    f3 Lret = mk3(0.f, 0.f, 0.f);
    enum { ENTER, ADVANCE, RETURN } state = ENTER;
    for (;;) {
        if (state == ENTER) {
            // a call at depth == sp
            RayCtx r;
            ray_setup(r, ro, rdir);
            HitRec h;
            ++nClosest;
            const bool hit = kd_trace<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, RT_INF, h, cn);
            Lret = mk3(0.f, 0.f, 0.f);
            state = RETURN;
            if (hit) {
                Surf sf;
                surface_at<SPH>(sc, h.prim, h.b0, h.b1, h.b2, ro, rdir, sf);
                const f3 wo = normalize3(-rdir);
                DBsdf bs;
                bsdf_init<MATSET_ANY>(bs, sf, sc.mats[sf.mat]);
                f3 L = mk3(0.f, 0.f, 0.f);
                if (!countOnly && sf.light >= 0) L = L + light_L(sc.lights[sf.light], sf.n, wo);
                for (uint32_t li = 0; li < sc.nLights; ++li) {
                    const float u0 = pcg_float(rng), u1 = pcg_float(rng);
                    nDraws += 2;
                    if (countOnly) continue;
                    f3 wi, pShape;
                    float pdf;
                    const f3 Li = light_sample_Li<SPH>(sc, sc.lights[li], sf.p, u0, u1, wi, pdf, pShape);
                    if (black3(Li) || pdf == 0) continue;
                    const f3 f = bsdf_f(bs, wo, wi, BSDF_ALL);
                    if (!black3(f)) {
                        RayCtx sr;
                        ray_setup(sr, sf.p, normalize3(pShape - sf.p));
                        HitRec sh;
                        ++nAny;
                        if (!kd_trace<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, sr, 1 - RT_SHADOW_EPS, sh, cn))
                            L = L + f * Li * absdot3(wi, sf.n) / pdf;
                    }
                }
                if (sp + 1 < maxDepth && sp < RT_WHITTED_MAX_FRAMES) {
                    WFrame &f = fr[sp++];
                    f.L = L;
                    f.prim = h.prim; f.b0 = h.b0; f.b1 = h.b1; f.b2 = h.b2;
                    f.ro = ro;
                    f.rd = rdir;
                    f.stage = 0;
                    f.pdf = 0.f;
                    state = ADVANCE;
                } else {
                    Lret = L;
                }
            }
        } else if (state == ADVANCE) {
            WFrame &f = fr[sp - 1];
            if (f.stage == 2) {
                Lret = f.L;
                --sp;
                state = RETURN;
                continue;
            }
            // stage 0: specularReflect, stage 1: specularTransmit
            Surf sf;
            surface_at<SPH>(sc, f.prim, f.b0, f.b1, f.b2, f.ro, f.rd, sf);
            DBsdf bs;
            bsdf_init<MATSET_ANY>(bs, sf, sc.mats[sf.mat]);
            const f3 wo = normalize3(-f.rd);
            const int type = (f.stage == 0 ? BSDF_REFLECTION : BSDF_TRANSMISSION) | BSDF_SPECULAR;
            f.stage += 1;
            f3 wi = mk3(0.f, 0.f, 0.f);
            float pdf = 0;
            int sampledType = 0;
            const float u0 = pcg_float(rng), u1 = pcg_float(rng);
            nDraws += 2;
            const f3 fv = bsdf_sample(bs, wo, wi, u0, u1, pdf, sampledType, type);
            if (pdf > 0.f && !black3(fv) && absdot3(wi, sf.n) != 0.f) {
                f.f = fv;
                f.absd = absdot3(wi, sf.n);
                f.pdf = pdf;
                ro = sf.p;
                rdir = normalize3(wi);
                state = ENTER;
            }
        } else {   // RETURN: a call finished with radiance Lret
            if (sp == 0) return Lret;
            WFrame &f = fr[sp - 1];
            f.L = f.L + f.f * Lret * f.absd / f.pdf;
            state = ADVANCE;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Path integrator on the reference's random stream (Render/Render.cpp:32-92, 348-449)
// ---------------------------------------------------------------------------------------------------
// PathRenderer::Li for one camera ray, start to finish in one lane, drawing from `rng` in the reference's order (light
// choice, light sample, scattering sample, BSDF sample, Russian roulette). The arithmetic is that of k_shade / k_shadow
// / k_probe; only here the rays of a path are traced where they arise, because the next sample's random numbers are
// the ones that follow this path's last draw.
template <bool LDS_SCENE, bool STATS, bool SPH>
RT_DEV f3 path_li(const DevScene &sc, const SceneTables &tb, uint32_t *stk, f3 ro, f3 rdir, Pcg &rng, int maxDepth, float rrThreshold,
                  TravCounters &cn, uint32_t &nClosest, uint32_t &nAny) {
    f3 L = mk3(0.f, 0.f, 0.f), beta = mk3(1.f, 1.f, 1.f);
    float etaScale = 1.f;
    bool specularBounce = false;
    for (int bounce = 0;; ++bounce) {
        RayCtx r;
        ray_setup(r, ro, rdir);
        HitRec h;
        ++nClosest;
        const bool hit = kd_trace<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, r, RT_INF, h, cn);
        if (!hit) break;
        Surf sf;
        surface_at<SPH>(sc, h.prim, h.b0, h.b1, h.b2, ro, rdir, sf);
        if ((bounce == 0 || specularBounce) && sf.light >= 0) L = L + beta * light_L(sc.lights[sf.light], sf.n, -rdir);
        if (bounce >= maxDepth) break;
        DBsdf bs;
        bsdf_init<MATSET_ANY>(bs, sf, sc.mats[sf.mat]);
        const f3 isectWo = normalize3(-rdir);
        if (bsdf_num(bs, BSDF_ALL & ~BSDF_SPECULAR) > 0 && sc.nLights > 0) {
            // uniformSampleOneLight -> estimateDirect (Render/Render.cpp:189-312)
            float selPdf;
            const int li = sample_light(sc, pcg_float(rng), selPdf);
            if (selPdf != 0) {
                const float ul0 = pcg_float(rng), ul1 = pcg_float(rng);
                const float us0 = pcg_float(rng), us1 = pcg_float(rng);
                const DevLight lt = sc.lights[li];
                const int flags = BSDF_ALL & ~BSDF_SPECULAR;
                f3 Ld = mk3(0.f, 0.f, 0.f);
                f3 wi, pShape;
                float lightPdf = 0, scatteringPdf = 0;
                const f3 Li = light_sample_Li<SPH>(sc, lt, sf.p, ul0, ul1, wi, lightPdf, pShape);
                if (lightPdf > 0 && !black3(Li)) {
                    const f3 f = bsdf_f(bs, isectWo, wi, flags) * absdot3(wi, sf.n);
                    scatteringPdf = bsdf_pdf(bs, isectWo, wi, flags);
                    if (!black3(f)) {
                        const float weight = power_heuristic(lightPdf, scatteringPdf);
                        const f3 A = f * Li * weight / lightPdf;
                        RayCtx sr;
                        ray_setup(sr, sf.p, normalize3(pShape - sf.p));
                        HitRec sh;
                        ++nAny;
                        if (!kd_trace<false, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, sr, 1 - RT_SHADOW_EPS, sh, cn))
                            Ld = Ld + A;
                    }
                }
                {
                    f3 wi2 = mk3(0.f, 0.f, 0.f);
                    float pdfB = scatteringPdf, weightB = 1;
                    int sampledType = 0;
                    f3 f2 = bsdf_sample(bs, isectWo, wi2, us0, us1, pdfB, sampledType, flags);
                    f2 = f2 * absdot3(wi2, sf.n);
                    if (!black3(f2) && pdfB > 0) {
                        bool go = true;
                        if ((sampledType & BSDF_SPECULAR) == 0) {
                            const float lp = light_pdf_Li<SPH>(sc, lt, sf.p, wi2);
                            if (lp == 0) go = false;
                            else weightB = power_heuristic(pdfB, lp);
                        }
                        if (go) {
                            RayCtx pr;
                            ray_setup(pr, sf.p, normalize3(wi2));
                            HitRec ph;
                            ++nClosest;
                            f3 Li2 = mk3(0.f, 0.f, 0.f);
                            if (kd_trace<true, STATS, StackCfg<LDS_SCENE>::LDS_N, LDS_SCENE, SPH>(sc, tb, stk, threadIdx.x, pr, RT_INF, ph, cn)) {
                                Surf ps;
                                surface_at<SPH>(sc, ph.prim, ph.b0, ph.b1, ph.b2, sf.p, pr.d, ps);
                                if (ps.light == li) Li2 = light_L(sc.lights[li], ps.n, -wi2);
                            }
                            if (!black3(Li2)) Ld = Ld + f2 * Li2 * mk3(1.f, 1.f, 1.f) * weightB / pdfB;
                        }
                    }
                }
                L = L + beta * (Ld / selPdf);
            }
        }
        // next direction
        const f3 wo = -rdir;
        f3 wi = mk3(0.f, 0.f, 0.f);
        float pdf = 0;
        int flags = 0;
        const float u0 = pcg_float(rng), u1 = pcg_float(rng);
        const f3 f = bsdf_sample(bs, wo, wi, u0, u1, pdf, flags, BSDF_ALL);
        if (black3(f) || pdf == 0.f) break;
        beta = beta * (f * absdot3(wi, sf.n) / pdf);
        specularBounce = (flags & BSDF_SPECULAR) != 0;
        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
            const float eta = bs.eta;
            etaScale *= (dot3(wo, sf.n) > 0) ? (eta * eta) : 1 / (eta * eta);
        }
        ro = sf.p;
        rdir = normalize3(wi);
        const f3 rrBeta = beta * etaScale;
        const float mx = gmax(gmax(rrBeta.x, rrBeta.y), rrBeta.z);
        if (mx < rrThreshold && bounce > 3) {
            const float q = gmax(.05f, 1 - mx);
            if (pcg_float(rng) < q) break;
            beta = beta / (1 - q);
        }
    }
    return L;
}

// The reference's render loop for the Path integrator with its own random stream: tile t draws from PCG32 sequence t,
// pixel after pixel, sample after sample (Render/Render.cpp:32-92). The stream is serial inside a tile, so a tile is
// one lane's work; each tile gets a wave to itself (lane 0), which keeps 1024 SIMDs busy with thousands of tiles and
// costs no divergence. This mode exists to reproduce the CPU renderer's image exactly; RT_RNG_COUNTER is the fast one.
template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_path_tile_stream(DevScene sc, CameraDev cam, int spp, int maxDepth, float rrThreshold, uint32_t nTiles,
                                                               uint32_t ntx, float *rgbSum, float *weightSum, float *spillRgb, float *spillW,
                                                               unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem);
    TravCounters cn = {0, 0, 0, 0, 0};
    uint32_t nc = 0, na = 0, ns = 0;
    const uint32_t wavesPerBlock = RT_BLOCK / 64;
    const uint32_t t = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0 && t < nTiles) {
        const int tx = (int)(t % ntx), ty = (int)(t / ntx);
        const int x0 = cam.sb[0] + tx * 16, x1 = min(x0 + 16, cam.sb[2]);
        const int y0 = cam.sb[1] + ty * 16, y1 = min(y0 + 16, cam.sb[3]);
        Pcg rng;
        pcg_set_sequence(rng, (uint64_t)t);
        for (int py = y0; py < y1; ++py)
            for (int px = x0; px < x1; ++px) {
                const bool onFilm = px >= cam.cb[0] && px < cam.cb[2] && py >= cam.cb[1] && py < cam.cb[3];
                const size_t g = onFilm ? (size_t)py * cam.width + px : 0;
                f3 acc = mk3(0.f, 0.f, 0.f);
                float w = 0.f;
                if (onFilm) {
                    acc = mk3(rgbSum[3 * g], rgbSum[3 * g + 1], rgbSum[3 * g + 2]);
                    w = weightSum[g];
                }
                for (int s = 0; s < spp; ++s) {
                    const float u0 = pcg_float(rng), u1 = pcg_float(rng);
                    const float pfx = (float)px + u0, pfy = (float)py + u1;
                    f3 o, d;
                    camera_ray(cam, pfx, pfy, o, d);
                    f3 L = path_li<LDS_SCENE, STATS, SPH>(sc, tb, stk, o, d, rng, maxDepth, rrThreshold, cn, nc, na);
                    L = film_clamp(cam, sanitize_L(L));
                    ++ns;
                    int fx0, fy0, fx1, fy1;
                    if (film_footprint(cam, pfx, pfy, px, py, fx0, fy0, fx1, fy1)) {
                        acc = acc + L;
                        w += 1.0f;
                    } else {
                        film_spill(cam, spillRgb, spillW, px, py, fx0, fy0, fx1, fy1, L, acc, w);
                    }
                }
                if (onFilm) {
                    rgbSum[3 * g] = acc.x; rgbSum[3 * g + 1] = acc.y; rgbSum[3 * g + 2] = acc.z;
                    weightSum[g] = w;
                }
            }
    }
    if (STATS) flush_counters(stats, cn, nc + na);
    if (nc) atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)nc);
    if (na) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)na);
    if (ns) atomicAdd(&stats[ST_CAMERA_SAMPLES], (unsigned long long)ns);
}

// Reference tile stream, step 1: one lane per 16x16 tile walks the tile's samples in the reference's
// order (Render/Render.cpp:32-92) consuming random numbers exactly as the full evaluation would, and
// records the PCG32 state at the first sample of every pixel.
template <bool LDS_SCENE, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_whitted_prepass(DevScene sc, CameraDev cam, int spp, int maxDepth, uint32_t nTiles, uint32_t ntx,
                                                              uint2 *pixelState) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem);
    const uint32_t t = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (t >= nTiles) return;
    const int tx = (int)(t % ntx), ty = (int)(t / ntx);
    const int x0 = cam.sb[0] + tx * 16, x1 = min(x0 + 16, cam.sb[2]);
    const int y0 = cam.sb[1] + ty * 16, y1 = min(y0 + 16, cam.sb[3]);
    const int sbW = cam.sb[2] - cam.sb[0];
    Pcg rng;
    pcg_set_sequence(rng, (uint64_t)t);
    TravCounters cn = {0, 0, 0, 0};
    uint32_t nc = 0, na = 0, nd = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            pixelState[(size_t)(y - cam.sb[1]) * sbW + (x - cam.sb[0])] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
            for (int s = 0; s < spp; ++s) {
                const float u0 = pcg_float(rng), u1 = pcg_float(rng);
                f3 o, d;
                camera_ray(cam, (float)x + u0, (float)y + u1, o, d);
                whitted_li<LDS_SCENE, false, SPH>(sc, tb, stk, o, d, rng, maxDepth, true, cn, nc, na, nd);
            }
        }
}

// Step 2 (and the whole job in counter mode): one lane per pixel evaluates its samples in order and adds
// them to the film.
// Reference tile stream, parallel form of step 1. A pixel's samples start `offset` draws into its tile's stream,
// where offset = draws of all earlier pixels of the tile - which depend on where THEIR samples fall, i.e. on their own
// offsets. Fixed point: every pixel counts the draws of its samples from a guessed offset (one lane per pixel, all
// tiles at once), a scan turns the counts into new offsets, and the two steps repeat until no offset changes. The
// count of a pixel rarely depends on the jitter (all its samples hit or all miss), so a good first guess converges
// at once; after pass i the first i pixels of every tile are final, so 256 passes always suffice, and a tile whose
// offsets a scan left unchanged is final and is skipped from then on (tileLive). The serial kernel above remains the
// fallback. Slot = tile * 256 + row-major position in the 16x16 tile.
template <bool LDS_SCENE, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_whitted_count(DevScene sc, CameraDev cam, int spp, int maxDepth, uint32_t nTiles, uint32_t ntx,
                                                            const unsigned long long *__restrict__ offsets, const uint32_t *__restrict__ tileLive,
                                                            unsigned long long *__restrict__ draws, uint2 *pixelState) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem);
    const int sbW = cam.sb[2] - cam.sb[0], sbH = cam.sb[3] - cam.sb[1];
    const uint32_t nSlots = nTiles * 256u;
    TravCounters cn = {0, 0, 0, 0, 0};
    uint32_t nc = 0, na = 0;
    for (uint32_t pi = blockIdx.x * RT_BLOCK + threadIdx.x; pi < nSlots; pi += gridDim.x * RT_BLOCK) {
        const uint32_t tile = pi / 256, inTile = pi % 256;
        if (!tileLive[tile]) continue;
        const int lx = (int)(tile % ntx) * 16 + (int)(inTile % 16), ly = (int)(tile / ntx) * 16 + (int)(inTile / 16);
        if (lx >= sbW || ly >= sbH) { draws[pi] = 0; continue; }
        Pcg rng;
        pcg_set_sequence(rng, (uint64_t)tile);
        pcg_advance(rng, offsets[pi]);
        pixelState[(size_t)ly * sbW + lx] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
        const int x = cam.sb[0] + lx, y = cam.sb[1] + ly;
        unsigned long long total = 0;
        for (int s = 0; s < spp; ++s) {
            const float u0 = pcg_float(rng), u1 = pcg_float(rng);
            f3 o, d;
            camera_ray(cam, (float)x + u0, (float)y + u1, o, d);
            uint32_t nd = 2;
            whitted_li<LDS_SCENE, false, SPH>(sc, tb, stk, o, d, rng, maxDepth, true, cn, nc, na, nd);
            total += nd;
        }
        draws[pi] = total;
    }
}

// One lane per tile: exclusive scan of the tile's per-pixel draw counts into offsets; flags any change.
// firstGuess != 0: instead of reading counts, assume every pixel of the film draws firstGuess numbers.
__global__ __launch_bounds__(RT_BLOCK) void k_whitted_scan(CameraDev cam, uint32_t nTiles, uint32_t ntx, const unsigned long long *__restrict__ draws,
                                                           unsigned long long *__restrict__ offsets, unsigned long long firstGuess, uint32_t *tileLive,
                                                           uint32_t *changed) {
    const uint32_t t = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (t >= nTiles) return;
    if (!firstGuess && !tileLive[t]) return;
    const int sbW = cam.sb[2] - cam.sb[0], sbH = cam.sb[3] - cam.sb[1];
    unsigned long long run = 0;
    bool diff = false;
    for (uint32_t k = 0; k < 256; ++k) {
        const size_t pi = (size_t)t * 256 + k;
        unsigned long long dcount;
        if (firstGuess) {
            const int lx = (int)(t % ntx) * 16 + (int)(k % 16), ly = (int)(t / ntx) * 16 + (int)(k / 16);
            dcount = (lx < sbW && ly < sbH) ? firstGuess : 0ull;
        } else {
            dcount = draws[pi];
        }
        if (firstGuess || offsets[pi] != run) {
            offsets[pi] = run;
            diff = true;
        }
        run += dcount;
    }
    tileLive[t] = diff ? 1u : 0u;
    if (diff && !firstGuess) atomicOr(changed, 1u);
}

template <bool LDS_SCENE, bool STATS, bool SPH>
__global__ __launch_bounds__(RT_BLOCK) void k_whitted(DevScene sc, CameraDev cam, int sppBegin, int sppEnd, int sppTotal, int maxDepth, int rngMode,
                                                      uint64_t seed, const uint2 *pixelState, float *rgbSum, float *weightSum, float *spillRgb,
                                                      float *spillW, unsigned long long *stats) {
    RT_DYNAMIC_LDS(smem);
    SceneTables tb;
    tables_setup<LDS_SCENE>(sc, smem, tb);
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem);
    const int sbW = cam.sb[2] - cam.sb[0], sbH = cam.sb[3] - cam.sb[1];
    TravCounters cn = {0, 0, 0, 0};
    uint32_t nc = 0, na = 0, ns = 0, ndr = 0;
    // pixels are visited tile by tile (16x16) so that a wave works on a compact screen region
    const uint32_t ntx = (uint32_t)(sbW + 15) / 16, nty = (uint32_t)(sbH + 15) / 16;
    const uint32_t nSlots = ntx * nty * 256u;
    for (uint32_t pi = blockIdx.x * RT_BLOCK + threadIdx.x; pi < nSlots; pi += gridDim.x * RT_BLOCK) {
        const uint32_t tile = pi / 256, inTile = pi % 256;
        const int lx = (int)(tile % ntx) * 16 + (int)(inTile % 16), ly = (int)(tile / ntx) * 16 + (int)(inTile / 16);
        if (lx >= sbW || ly >= sbH) continue;
        const int px = cam.sb[0] + lx, py = cam.sb[1] + ly;
        const bool onFilm = px >= cam.cb[0] && px < cam.cb[2] && py >= cam.cb[1] && py < cam.cb[3];
        const size_t g = onFilm ? (size_t)py * cam.width + px : 0;
        f3 acc = mk3(0.f, 0.f, 0.f);
        float w = 0.f;
        if (onFilm) {
            acc = mk3(rgbSum[3 * g], rgbSum[3 * g + 1], rgbSum[3 * g + 2]);
            w = weightSum[g];
        }
        Pcg rng;
        if (rngMode == RT_RNG_REFERENCE_TILE_STREAM) {
            const uint32_t tIdx = (uint32_t)(ly / 16) * ntx + (uint32_t)(lx / 16);
            rng.inc = ((uint64_t)tIdx << 1u) | 1u;
            const uint2 st = pixelState[(size_t)ly * sbW + lx];
            rng.state = (uint64_t)st.x | ((uint64_t)st.y << 32);
        }
        for (int s = sppBegin; s < sppEnd; ++s) {
            if (rngMode == RT_RNG_COUNTER)
                pcg_set_sequence(rng, counter_sequence((uint64_t)py * (uint64_t)cam.width + (uint64_t)px, (uint32_t)sppTotal, (uint32_t)s, seed));
            const float u0 = pcg_float(rng), u1 = pcg_float(rng);
            const float pfx = (float)px + u0, pfy = (float)py + u1;
            f3 o, d;
            camera_ray(cam, pfx, pfy, o, d);
            f3 L = whitted_li<LDS_SCENE, STATS, SPH>(sc, tb, stk, o, d, rng, maxDepth, false, cn, nc, na, ndr);
            L = film_clamp(cam, sanitize_L(L));
            ++ns;
            int fx0, fy0, fx1, fy1;
            if (film_footprint(cam, pfx, pfy, px, py, fx0, fy0, fx1, fy1)) {
                acc = acc + L;
                w += 1.0f;
            } else {
                film_spill(cam, spillRgb, spillW, px, py, fx0, fy0, fx1, fy1, L, acc, w);
            }
        }
        if (onFilm) {
            rgbSum[3 * g] = acc.x; rgbSum[3 * g + 1] = acc.y; rgbSum[3 * g + 2] = acc.z;
            weightSum[g] = w;
        }
    }
    if (STATS) flush_counters(stats, cn, nc + na);
    if (nc) atomicAdd(&stats[ST_RAYS_CLOSEST], (unsigned long long)nc);
    if (na) atomicAdd(&stats[ST_RAYS_ANY], (unsigned long long)na);
    if (ns) atomicAdd(&stats[ST_CAMERA_SAMPLES], (unsigned long long)ns);
}
