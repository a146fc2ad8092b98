// C ABI of the device library (include/raytracing_amd.h): context, scene upload, batch tracing, rendering.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library is opened at run time by rt_create_multi (dlopen), never linked
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <limits>
#include <vector>

#include "../../../include/raytracing_amd.h"
#include "kernels.h"

#ifndef RT_PIPES
#define RT_PIPES RT_MAX_PIPELINES   // wavefront pipelines (streams + buffer sets) that overlap consecutive passes: 2 -> 3 gains 2 %, 4 loses 4 %
#endif

// ---------------------------------------------------------------------------------------------------
// Scratch of rt_render_noalloc_dev: one array per GPU, part of the code object, so the entry point never allocates.
// Layout (byte offsets, all multiples of 256):
//   [statistics counters][queue counters][film spill rgb][film spill weight][one wavefront pipeline's queues]
// Host code never uses the array's name as an address: rt_create asks the runtime where the array lives on its device
// (hipGetSymbolAddress) and everything below goes through that pointer.
// ---------------------------------------------------------------------------------------------------
namespace noalloc {
constexpr size_t kMaxPaths = RT_NOALLOC_MAX_PATHS;
constexpr size_t kMaxDepth = RT_NOALLOC_MAX_DEPTH;
constexpr size_t align256(size_t n) { return (n + 255) / 256 * 256; }
constexpr size_t kStatsOff = 0;
constexpr size_t kCountsOff = kStatsOff + align256(ST_COUNT * sizeof(unsigned long long));
constexpr size_t kCountsBytes = (kMaxDepth + 2) * RT_CNT_PER_BOUNCE * RT_CNT_STRIDE * sizeof(uint32_t);
constexpr size_t kSpillRgbOff = kCountsOff + align256(kCountsBytes);
constexpr size_t kSpillWOff = kSpillRgbOff + align256(kMaxPaths * 3 * sizeof(float));
constexpr size_t kQueuesOff = kSpillWOff + align256(kMaxPaths * sizeof(float));
constexpr size_t kQueueVecPerPath = 11 + 6;   // PathBuffers: 11 float4 arrays + the 6-float4 probe record
constexpr size_t kBytes = kQueuesOff + kMaxPaths * kQueueVecPerPath * 16;
constexpr int kMaxDevices = 64;
}  // namespace noalloc

static_assert(noalloc::kBytes <= ((size_t)RT_NOALLOC_SCRATCH_MB << 20), "the header states the size of the scratch array");
static __device__ __align__(256) unsigned char scratch[noalloc::kBytes];

namespace {

std::string g_create_err;
std::mutex g_scratchMutex;
unsigned char *g_scratchPtr[noalloc::kMaxDevices] = {};   // device address of `scratch` on each GPU, set by rt_create

// The calling thread's current device must be `device`.
hipError_t scratch_address(int device, unsigned char **out) {
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (device < 0 || device >= noalloc::kMaxDevices) return hipErrorInvalidDevice;
    if (!g_scratchPtr[device]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratchPtr[device] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}

struct EventPair {
    hipEvent_t a, b;
};

}  // namespace

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int numCUs = 256;
    int gridBlocks = 2048;
    std::map<const void *, int> residentGrid;   // streaming kernels: blocks that are resident at once, per kernel

    // scene
    bool haveScene = false;
    DevScene sc{};
    void *dNodes = nullptr, *dLeafIdx = nullptr, *dTriHot = nullptr, *dTriShade = nullptr, *dMats = nullptr, *dLights = nullptr;
    void *dLightCdf = nullptr, *dLightFunc = nullptr, *dSpheres = nullptr;
    unsigned long long *dWhittedOffsets = nullptr, *dWhittedDraws = nullptr;   // tile-stream prepass scratch
    size_t whittedScanCap = 0;
    int whittedPasses = 0;   // count passes the last tile-stream prepass took (-1: fell back to the serial kernel)
    bool ldsScene = false;
    bool hasSpheres = false;
    bool hasGlass = false;
    bool allLambert = false;

    // camera
    bool haveCamera = false;
    CameraDev cam{};

    // stats
    unsigned long long *dStats = nullptr;
    int traceSchedule = RT_SCHEDULE_AUTO;
    size_t shadeLdsBytes = 0;   // dynamic LDS of k_shade when it shades out of a copy of the tables in LDS; 0 = tables in memory
    // Scenes in LDS: first bounce whose extend stage runs the refilling kernel. Camera rays are coherent and stay with the
    // lockstep kernel (11.98 against 11.74 G rays/s with the refilling kernel from bounce 0, 11.57 from bounce 2); the shadow
    // stage stays lockstep too: its rays are 0.9999 long, their cost is the set-up, which lockstep runs at full width
    // (11.98 against 11.30 G rays/s with refilling shadow kernels).
    int refillExtendFrom = 1;
    int pipelines = RT_PIPES;   // RT_OPTION_PIPELINES
    bool kernelTimes = false;   // RT_OPTION_KERNEL_TIMES: HIP events around the extend and shade launches of a Path render
    std::vector<EventPair> extendEvents, shadeEvents;   // HIP events around every extend / shade launch of the path tracer
    size_t extendEventsUsed = 0, shadeEventsUsed = 0;
    double renderMs = 0;

    // path tracer pipelines: consecutive passes rotate through kPipes sets of buffers, each on its own stream, so the
    // small late-bounce launches of one pass overlap the large early-bounce launches of the next ones
    static constexpr int kPipes = RT_PIPES;
    PathBuffers pb[kPipes]{};
    hipStream_t pipeStream[kPipes] = {};
    hipEvent_t pipeFilmDone[kPipes] = {};
    hipEvent_t mainReady = nullptr;
    size_t pbCapacity = 0;
    size_t pbBytesPerPipe = 0;
    int pbDepth = 0;
    std::vector<void *> pbAllocs;

    // film spill + whitted state
    float *dSpillRgb = nullptr, *dSpillW = nullptr;
    size_t spillPixels = 0;
    // multi-GPU group (rt_create_multi): this context is the group's device 0, `peers` are the other devices'
    // contexts. Scene, camera, film and options are mirrored to every member; a render is split by sample index and
    // the members' films are summed into this device's with one RCCL reduce.
    std::vector<rt_context *> peers;
    rt_context *groupRoot = nullptr;        // set in a peer: the context that owns it
    std::vector<ncclComm_t> comms;          // one per member, in member order; empty when the members share one GPU
    float *dGroupFilm = nullptr;            // packed film of this member: rgb sums (3 n) then weight sums (n)
    size_t groupFilmPixels = 0;
    hipEvent_t groupDone = nullptr;
    unsigned char *scratchPtr = nullptr;   // this GPU's copy of the static scratch array (rt_render_noalloc_dev)
    bool spillDirty = false;   // the last render returned an error: the spill buffers may hold its samples
    uint2 *dPixelState = nullptr;
    size_t pixelStateCap = 0;
};

namespace {

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return RT_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

int fail(rt_context *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

void free_scene(rt_context *c) {
    void **ptrs[] = {&c->dNodes, &c->dLeafIdx, &c->dTriHot, &c->dTriShade, &c->dMats, &c->dLights, &c->dLightCdf, &c->dLightFunc, &c->dSpheres};
    for (void **p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->haveScene = false;
}

void free_path_buffers(rt_context *c) {
    for (void *p : c->pbAllocs) (void)hipFree(p);
    c->pbAllocs.clear();
    c->pbCapacity = 0;
    c->pbBytesPerPipe = 0;
    c->pbDepth = 0;
    for (auto &pb : c->pb) pb = PathBuffers{};
}

// Per-triangle constants of the surface part of ATriangleShape::hit (reference
// src/core/Shape/TriangleShape.cpp:387-423,433): ss = normalize(dpdu), the geometric normal, and whether the
// closest-hit flavour rejects the triangle as degenerate. They do not depend on the ray.
void tri_constants(const float *p, const float *uvIn, f3 &ss, f3 &ng, bool &degenerate) {
    const f3 p0 = mk3(p[0], p[1], p[2]), p1 = mk3(p[3], p[4], p[5]), p2 = mk3(p[6], p[7], p[8]);
    float uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};
    if (uvIn)
        for (int k = 0; k < 3; ++k) { uv[k][0] = uvIn[2 * k]; uv[k][1] = uvIn[2 * k + 1]; }
    const float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
    const f3 dp02 = p0 - p2, dp12 = p1 - p2;
    const float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    const bool degenerateUV = std::fabs(determinant) < 1e-8;
    f3 dpdu = mk3(0, 0, 0), dpdv = mk3(0, 0, 0);
    if (!degenerateUV) {
        const float invdet = 1 / determinant;
        dpdu = (duv12[1] * dp02 - duv02[1] * dp12) * invdet;
        dpdv = (-duv12[0] * dp02 + duv02[0] * dp12) * invdet;
    }
    degenerate = false;
    if (degenerateUV || dot3(cross3(dpdu, dpdv), cross3(dpdu, dpdv)) == 0) {
        const f3 g = cross3(p2 - p0, p1 - p0);
        if (dot3(g, g) == 0) {
            degenerate = true;
        } else {
            const f3 v1 = normalize3(g);   // coordinateSystem (Utils/Math.h:454-461)
            if (std::fabs(v1.x) > std::fabs(v1.y)) dpdu = mk3(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
            else dpdu = mk3(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
        }
    }
    ss = normalize3(dpdu);
    ng = normalize3(cross3(dp02, dp12));
}

template <typename T>
int upload(rt_context *c, void **dst, const std::vector<T> &v, size_t padBytes = 16) {
    size_t bytes = v.size() * sizeof(T);
    size_t alloc = ((bytes + padBytes - 1) / padBytes) * padBytes + padBytes;
    HIPCHK(c, hipMalloc(dst, alloc));
    // The context's streams do not synchronise with the null stream, on which hipMemset runs asynchronously: every fill
    // in this file goes to the context's own stream, followed by a wait where host copies or other streams come next.
    HIPCHK(c, hipMemsetAsync(*dst, 0, alloc, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bytes) HIPCHK(c, hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

// Kernel variant for this context's scene: tables in LDS or HBM, traversal counters on or off, spheres or not.
#define RT_PICK(K, c, stats)                                                                                                    \
    ((c)->hasSpheres ? ((c)->ldsScene ? ((stats) ? K<true, true, true> : K<true, false, true>)                                  \
                                      : ((stats) ? K<false, true, true> : K<false, false, true>))                               \
                     : ((c)->ldsScene ? ((stats) ? K<true, true, false> : K<true, false, false>)                                \
                                      : ((stats) ? K<false, true, false> : K<false, false, false>)))

size_t trace_lds_bytes(const rt_context *c) { return c->ldsScene ? stack_lds_bytes<true>() + c->sc.ldsSceneBytes : stack_lds_bytes<false>(); }
size_t stream_lds_bytes_of(const rt_context *c) { return c->ldsScene ? stream_lds_bytes<true>() + c->sc.ldsSceneBytes : stream_lds_bytes<false>(); }

int ensure_spill(rt_context *c) {
    const size_t n = (size_t)c->cam.width * c->cam.height;
    if (c->spillPixels == n && c->dSpillRgb) {
        if (c->spillDirty) {   // leftovers of a render that failed half way must not reach the next film
            HIPCHK(c, hipMemsetAsync(c->dSpillRgb, 0, n * 3 * sizeof(float), c->stream));
            HIPCHK(c, hipMemsetAsync(c->dSpillW, 0, n * sizeof(float), c->stream));
            c->spillDirty = false;
        }
        return RT_OK;
    }
    if (c->dSpillRgb) (void)hipFree(c->dSpillRgb);
    if (c->dSpillW) (void)hipFree(c->dSpillW);
    c->dSpillRgb = c->dSpillW = nullptr;
    c->spillPixels = 0;
    HIPCHK(c, hipMalloc((void **)&c->dSpillRgb, n * 3 * sizeof(float)));
    HIPCHK(c, hipMalloc((void **)&c->dSpillW, n * sizeof(float)));
    HIPCHK(c, hipMemsetAsync(c->dSpillRgb, 0, n * 3 * sizeof(float), c->stream));   // the renderers' streams follow c->stream
    HIPCHK(c, hipMemsetAsync(c->dSpillW, 0, n * sizeof(float), c->stream));
    c->spillPixels = n;
    c->spillDirty = false;
    return RT_OK;
}

int ensure_path_buffers(rt_context *c, size_t nPaths, int maxDepth) {
    if (c->pbCapacity >= nPaths && c->pbDepth >= maxDepth) return RT_OK;
    free_path_buffers(c);
    size_t total = 0;
    auto alloc = [&](void **p, size_t bytes) -> int {
        HIPCHK(c, hipMalloc(p, bytes));
        c->pbAllocs.push_back(*p);
        total += bytes;
        return RT_OK;
    };
    int rc;
    for (int k = 0; k < rt_context::kPipes; ++k) {
        PathBuffers &pb = c->pb[k];
        for (int q = 0; q < 2; ++q) {
            if ((rc = alloc((void **)&pb.rayA[q], nPaths * 16))) return rc;
            if ((rc = alloc((void **)&pb.rayB[q], nPaths * 16))) return rc;
        }
        if ((rc = alloc((void **)&pb.hit, nPaths * 16))) return rc;
        if ((rc = alloc((void **)&pb.L, nPaths * 16))) return rc;
        if ((rc = alloc((void **)&pb.beta, nPaths * 16))) return rc;
        if ((rc = alloc((void **)&pb.rng, nPaths * 16))) return rc;
        if ((rc = alloc((void **)&pb.shA, nPaths * 16))) return rc;
        if ((rc = alloc((void **)&pb.shB, nPaths * 16))) return rc;
        if ((rc = alloc((void **)&pb.shC, nPaths * 16))) return rc;
        if ((rc = alloc((void **)&pb.probe, nPaths * 16 * 6))) return rc;
        if ((rc = alloc((void **)&pb.counts, (size_t)(maxDepth + 2) * RT_CNT_PER_BOUNCE * RT_CNT_STRIDE * sizeof(uint32_t)))) return rc;
    }
    c->pbCapacity = nPaths;
    c->pbBytesPerPipe = total / rt_context::kPipes;
    c->pbDepth = maxDepth;
    return RT_OK;
}

EventPair *next_event_pair(std::vector<EventPair> &events, size_t &used) {
    if (used >= 200000) return nullptr;
    if (used == events.size()) {
        EventPair ep;
        if (hipEventCreate(&ep.a) != hipSuccess) return nullptr;
        if (hipEventCreate(&ep.b) != hipSuccess) { (void)hipEventDestroy(ep.a); return nullptr; }
        events.push_back(ep);
    }
    return &events[used++];
}

double elapsed_ms(const std::vector<EventPair> &events, size_t used) {
    double ms = 0;
    for (size_t i = 0; i < used; ++i) {
        float m = 0;
        if (hipEventElapsedTime(&m, events[i].a, events[i].b) == hipSuccess) ms += m;
    }
    return ms;
}

// Grid of a streaming kernel: the workgroups the device keeps resident at once, so that every wave runs from the
// first ray to the last and the static chunk -> wave assignment is balanced.
template <typename K>
int resident_grid(rt_context *c, K kernel, size_t ldsBytes) {
    const void *key = reinterpret_cast<const void *>(kernel);
    auto it = c->residentGrid.find(key);
    if (it != c->residentGrid.end()) return it->second;
    int perCU = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, RT_BLOCK, ldsBytes) != hipSuccess || perCU <= 0) perCU = 4;
    const int grid = c->numCUs * std::min(perCU, 8);
    c->residentGrid[key] = grid;
    return grid;
}

template <typename K, typename... Args>
void launch_lockstep(rt_context *c, K k, uint64_t n, Args... args) {
    const int grid = (int)std::min<uint64_t>((n + RT_BLOCK - 1) / RT_BLOCK, (uint64_t)c->gridBlocks);
    hipLaunchKernelGGL(k, dim3(grid), dim3(RT_BLOCK), trace_lds_bytes(c), c->stream, args...);
}

template <typename K, typename... Args>
void launch_stream(rt_context *c, K k, uint64_t n, Args... args) {
    const size_t lds = stream_lds_bytes_of(c);
    const int grid = (int)std::min<uint64_t>((n + RT_BLOCK - 1) / RT_BLOCK, (uint64_t)resident_grid(c, k, lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(RT_BLOCK), lds, c->stream, args...);
}

// Refilling kernels: the lockstep kernels' LDS layout, and a grid that is resident as a whole (their waves split the rays
// statically, as the streaming kernels do).
template <typename K, typename... Args>
void launch_refill(rt_context *c, K k, uint64_t n, Args... args) {
    const size_t lds = trace_lds_bytes(c);
    const int grid = (int)std::min<uint64_t>((n + RT_BLOCK - 1) / RT_BLOCK, (uint64_t)resident_grid(c, k, lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(RT_BLOCK), lds, c->stream, args...);
}

// The schedule in force: RT_SCHEDULE_AUTO is lockstep for scenes in LDS and refilling for scenes in HBM (measured on the
// 1 M-triangle soup and the 4 M-triangle mesh against streaming: camera rays 1.90 / 1.66 and 1.12 / 1.01 G rays/s, incoherent
// rays 0.74 / 0.73 and 1.55 / 1.31, path tracing 536 / 489 and 717 / 646 M samples/s).
int schedule_in_force(const rt_context *c) {
    if (c->traceSchedule != RT_SCHEDULE_AUTO) return c->traceSchedule;
    return c->ldsScene ? RT_SCHEDULE_LOCKSTEP : RT_SCHEDULE_REFILL;
}
bool use_stream_schedule(const rt_context *c) { return schedule_in_force(c) == RT_SCHEDULE_STREAM; }

// The streaming kernels index rays with 32 bits: larger batches are cut into several launches.
constexpr uint64_t kMaxRaysPerLaunch = 1ull << 31;

int trace_closest_dev(rt_context *c, uint64_t n, const float *o, const float *d, const float *tmax, int32_t *prim, float *t, float *bary,
                      bool stats) {
    for (uint64_t base = 0; base < n; base += kMaxRaysPerLaunch) {
        const uint64_t m = std::min<uint64_t>(kMaxRaysPerLaunch, n - base);
        const float *o_ = o + 3 * base, *d_ = d + 3 * base, *tm_ = tmax ? tmax + base : nullptr;
        int32_t *prim_ = prim ? prim + base : nullptr;
        float *t_ = t ? t + base : nullptr, *bary_ = bary ? bary + 3 * base : nullptr;
        if (schedule_in_force(c) == RT_SCHEDULE_REFILL) {
            launch_refill(c, RT_PICK(k_trace_closest_refill, c, stats), m, c->sc, m, o_, d_, tm_, prim_, t_, bary_, c->dStats);
        } else if (use_stream_schedule(c)) {
            launch_stream(c, RT_PICK(k_trace_closest_stream, c, stats), m, c->sc, m, o_, d_, tm_, prim_, t_, bary_, c->dStats);
        } else {
            launch_lockstep(c, RT_PICK(k_trace_closest, c, stats), m, c->sc, m, o_, d_, tm_, prim_, t_, bary_, c->dStats);
        }
        HIPCHK(c, hipGetLastError());
    }
    return RT_OK;
}

int trace_any_dev(rt_context *c, uint64_t n, const float *o, const float *d, const float *tmax, uint8_t *occ, bool stats) {
    for (uint64_t base = 0; base < n; base += kMaxRaysPerLaunch) {
        const uint64_t m = std::min<uint64_t>(kMaxRaysPerLaunch, n - base);
        const float *o_ = o + 3 * base, *d_ = d + 3 * base, *tm_ = tmax ? tmax + base : nullptr;
        if (schedule_in_force(c) == RT_SCHEDULE_REFILL) {
            launch_refill(c, RT_PICK(k_trace_any_refill, c, stats), m, c->sc, m, o_, d_, tm_, occ + base, c->dStats);
        } else if (use_stream_schedule(c)) {
            launch_stream(c, RT_PICK(k_trace_any_stream, c, stats), m, c->sc, m, o_, d_, tm_, occ + base, c->dStats);
        } else {
            launch_lockstep(c, RT_PICK(k_trace_any, c, stats), m, c->sc, m, o_, d_, tm_, occ + base, c->dStats);
        }
        HIPCHK(c, hipGetLastError());
    }
    return RT_OK;
}

// Where a render keeps what it needs beyond its inputs and outputs: the context's own allocations (rt_render*), or
// this GPU's static scratch array (rt_render_noalloc_dev).
struct RenderScratch {
    PathBuffers *pb;        // nPipes sets of wavefront queues
    int nPipes;
    size_t capacityPaths;   // paths each set holds
    float *spillRgb, *spillW;
    unsigned long long *stats;
    bool timeExtend;        // HIP events around the extend launches (the events are created on demand)
};

int check_render_args(rt_context *c, const rt_render_params *p) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!p) return fail(c, RT_ERR_INVALID_ARG, "params is NULL");
    if (!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    if (!c->haveCamera) return fail(c, RT_ERR_NO_CAMERA, "no camera set");
    if (p->integrator != RT_INTEGRATOR_PATH && p->integrator != RT_INTEGRATOR_WHITTED) return fail(c, RT_ERR_INVALID_ARG, "unknown integrator");
    if (p->spp_total <= 0 || p->spp_begin < 0 || p->spp_end > p->spp_total || p->spp_begin > p->spp_end)
        return fail(c, RT_ERR_INVALID_ARG, "bad sample range");
    if (p->max_depth < 0 || p->max_depth > 1024) return fail(c, RT_ERR_INVALID_ARG, "bad max_depth");
    if (p->rng_mode != RT_RNG_COUNTER && p->rng_mode != RT_RNG_REFERENCE_TILE_STREAM) return fail(c, RT_ERR_INVALID_ARG, "unknown rng_mode");
    if (p->rng_mode == RT_RNG_REFERENCE_TILE_STREAM) {
        if (p->spp_begin != 0 || p->spp_end != p->spp_total)
            return fail(c, RT_ERR_INVALID_ARG, "the reference tile stream cannot be split by sample index");
    }
    if (p->integrator == RT_INTEGRATOR_WHITTED && p->max_depth > RT_WHITTED_MAX_FRAMES)
        return fail(c, RT_ERR_UNSUPPORTED, "Whitted max_depth above 32 is not supported");
    return RT_OK;
}

int render_path(rt_context *c, const rt_render_params *p, float *rgb, float *wsum, const RenderScratch *fixed) {
    const CameraDev &cam = c->cam;
    const uint32_t sbW = (uint32_t)(cam.sb[2] - cam.sb[0]), sbH = (uint32_t)(cam.sb[3] - cam.sb[1]);
    const uint64_t nPix = (uint64_t)sbW * sbH;
    if (nPix == 0 || p->spp_end == p->spp_begin) return RT_OK;
    int sppPass = p->spp_per_pass;
    if (sppPass <= 0) {
        // Up to 134 M paths in flight per pipeline (36 GB of queues): the late bounces of a pass are a dozen launches that a
        // few thousand paths keep busy for their latency, and bigger passes have fewer of them per sample (1080p: 13.2 G
        // rays/s with passes of 32 spp, 13.8 with 64, 13.9 with 96). The job is cut into equal passes, as many as fill every
        // pipeline in every round; a job that fits one pass is still cut in three when it is big enough to gain from overlap.
        const uint64_t job = (uint64_t)(p->spp_end - p->spp_begin);
        const uint64_t maxSpp = std::max<uint64_t>(1, (1ull << 27) / nPix);
        const uint64_t pipes = fixed ? 1u : (uint64_t)std::max(1, c->pipelines);
        uint64_t nPass = (job + maxSpp - 1) / maxSpp;
        if (nPass > 1 || (job >= pipes && job * nPix >= (1ull << 23))) nPass = (nPass + pipes - 1) / pipes * pipes;
        sppPass = (int)((job + nPass - 1) / nPass);
    }
    sppPass = std::min(sppPass, p->spp_end - p->spp_begin);
    if (nPix * (uint64_t)sppPass >= (1ull << 30)) sppPass = (int)std::max<uint64_t>(1, ((1ull << 30) - 1) / nPix);
    if (nPix >= (1ull << 30)) return fail(c, RT_ERR_UNSUPPORTED, "film too large");
    int rc;
    RenderScratch own;
    if (fixed) {
        // a fixed set of queues: as many samples per pixel in flight as fit
        if (nPix > fixed->capacityPaths) return fail(c, RT_ERR_UNSUPPORTED, "sample bounds hold more pixels than the scratch holds paths");
        sppPass = (int)std::min<uint64_t>((uint64_t)sppPass, fixed->capacityPaths / nPix);
    } else {
        // the queues are sized for a 288 GB part; on a device with less free memory fall back to fewer samples in flight
        while ((rc = ensure_path_buffers(c, nPix * (uint64_t)sppPass, p->max_depth)) != RT_OK) {
            free_path_buffers(c);
            (void)hipGetLastError();
            if (sppPass == 1 || p->spp_per_pass > 0) return rc;   // an explicit request is not second-guessed
            sppPass = (sppPass + 1) / 2;
        }
        if ((rc = ensure_spill(c))) return rc;
        own = RenderScratch{c->pb, rt_context::kPipes, c->pbCapacity, c->dSpillRgb, c->dSpillW, c->dStats, c->kernelTimes};
    }
    const RenderScratch &rs = fixed ? *fixed : own;
    const bool stats = p->collect_stats != 0;
    const size_t lds = trace_lds_bytes(c), ldsStream = stream_lds_bytes_of(c);
    const int grid = c->gridBlocks;
    // scenes in HBM: the extend and shadow kernels regroup rays inside the wave (dstream.h). Their static work split is
    // balanced when the grid is resident as a whole, which kernels of overlapping passes take away from each other; the
    // overlap still wins (1080p, passes of 32 spp: +10 % on the 1 M-triangle soup, +8 % on the 4 M-triangle mesh), because
    // the late bounces of a pass are a few latency-bound launches that leave the device idle
    const bool streamTrace = use_stream_schedule(c);
    const bool refillTrace = schedule_in_force(c) == RT_SCHEDULE_REFILL && !c->ldsScene;   // scenes in HBM: every bounce, and the shadow stage
    const int pipesInUse = std::min(c->pipelines, rs.nPipes);

    // what camera_origin() returns on the device: xf_apply(c2w, (0, 0, 0), 1) with the same operation order
    float co[4];
    for (int k = 0; k < 4; ++k) co[k] = (cam.c2w[0 * 4 + k] * 0.f + cam.c2w[1 * 4 + k] * 0.f) + (cam.c2w[2 * 4 + k] * 0.f + cam.c2w[3 * 4 + k] * 1.0f);
    if (co[3] != 1) for (int k = 0; k < 3; ++k) co[k] = co[k] / co[3];
    const float4 camOrigin = make_float4(co[0], co[1], co[2], 0.f);

    // the pipelines start after everything already queued on the context's stream
    HIPCHK(c, hipEventRecord(c->mainReady, c->stream));
    for (int k = 0; k < rt_context::kPipes; ++k) HIPCHK(c, hipStreamWaitEvent(c->pipeStream[k], c->mainReady, 0));
    int lastPipe = -1;
    int passIndex = 0;
    for (int s0 = p->spp_begin; s0 < p->spp_end; s0 += sppPass, ++passIndex) {
        const int k = passIndex % pipesInUse;
        PathBuffers &pb = rs.pb[k];
        hipStream_t st = c->pipeStream[k];
        PassInfo ps;
        ps.nPix = (uint32_t)nPix;
        ps.sbW = sbW;
        ps.sppPass = (uint32_t)std::min(sppPass, p->spp_end - s0);
        ps.sppBase = (uint32_t)s0;
        ps.sppTotal = (uint32_t)p->spp_total;
        ps.seed = p->seed;
        ps.maxDepth = p->max_depth;
        ps.rrThreshold = p->rr_threshold;
        HIPCHK(c, hipMemsetAsync(pb.counts, 0, (size_t)(p->max_depth + 2) * RT_CNT_PER_BOUNCE * RT_CNT_STRIDE * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(RT_BLOCK), 0, st, cam, ps, pb, rs.stats);
        for (int b = 0; b <= p->max_depth; ++b) {
            const int cur = b & 1;
            // camera rays share their origin and have no origin records
            auto originsOf = [&](int bounce) { return bounce == 0 ? (const float4 *)nullptr : (const float4 *)pb.rayA[cur]; };
            EventPair *ep = rs.timeExtend ? next_event_pair(c->extendEvents, c->extendEventsUsed) : nullptr;
            if (ep) HIPCHK(c, hipEventRecord(ep->a, st));
            if (streamTrace) {
                auto kx = RT_PICK(k_extend_stream, c, stats);
                hipLaunchKernelGGL(kx, dim3(resident_grid(c, kx, ldsStream)), dim3(RT_BLOCK), ldsStream, st, c->sc, originsOf(b),
                                   (const float4 *)pb.rayB[cur], pb.hit, (const uint32_t *)&pb.counts[cnt_index(b, 0)], rs.stats, camOrigin);
            } else if (refillTrace || (c->ldsScene && b >= c->refillExtendFrom)) {
                auto kx = RT_PICK(k_extend_refill, c, stats);
                hipLaunchKernelGGL(kx, dim3(resident_grid(c, kx, lds)), dim3(RT_BLOCK), lds, st, c->sc, originsOf(b), (const float4 *)pb.rayB[cur], pb.hit,
                                   (const uint32_t *)&pb.counts[cnt_index(b, 0)], rs.stats, camOrigin);
            } else {
                auto kx = RT_PICK(k_extend, c, stats);
                hipLaunchKernelGGL(kx, dim3(grid), dim3(RT_BLOCK), lds, st, c->sc, originsOf(b), (const float4 *)pb.rayB[cur], pb.hit,
                                   (const uint32_t *)&pb.counts[cnt_index(b, 0)], rs.stats, camOrigin);
            }
            if (ep) HIPCHK(c, hipEventRecord(ep->b, st));
            auto ksh = c->shadeLdsBytes ? (c->hasSpheres ? (c->allLambert ? k_shade<MATSET_LAMBERT, true, true> : k_shade<MATSET_ANY, true, true>)
                                                         : (c->allLambert ? k_shade<MATSET_LAMBERT, false, true> : k_shade<MATSET_ANY, false, true>))
                                        : (c->hasSpheres ? (c->allLambert ? k_shade<MATSET_LAMBERT, true, false> : k_shade<MATSET_ANY, true, false>)
                                                         : (c->allLambert ? k_shade<MATSET_LAMBERT, false, false> : k_shade<MATSET_ANY, false, false>));
            EventPair *es = rs.timeExtend ? next_event_pair(c->shadeEvents, c->shadeEventsUsed) : nullptr;
            if (es) HIPCHK(c, hipEventRecord(es->a, st));
            hipLaunchKernelGGL(ksh, dim3(grid), dim3(RT_BLOCK), c->shadeLdsBytes, st, c->sc, cam, ps, pb, b);
            if (es) HIPCHK(c, hipEventRecord(es->b, st));
            if (b < p->max_depth) {
                if (streamTrace) {
                    auto ks = RT_PICK(k_shadow_stream, c, stats);
                    hipLaunchKernelGGL(ks, dim3(resident_grid(c, ks, ldsStream)), dim3(RT_BLOCK), ldsStream, st, c->sc, pb, b, rs.stats);
                } else if (refillTrace) {
                    auto ks = RT_PICK(k_shadow_refill, c, stats);
                    hipLaunchKernelGGL(ks, dim3(resident_grid(c, ks, lds)), dim3(RT_BLOCK), lds, st, c->sc, pb, b, rs.stats);
                } else {
                    auto ks = RT_PICK(k_shadow, c, stats);
                    hipLaunchKernelGGL(ks, dim3(grid), dim3(RT_BLOCK), lds, st, c->sc, pb, b, rs.stats);
                }
                auto kp = RT_PICK(k_probe, c, stats);
                hipLaunchKernelGGL(kp, dim3(grid), dim3(RT_BLOCK), lds, st, c->sc, pb, b, rs.stats);
            }
        }
        // film accumulation is ordered across passes: pass i+1 adds its samples after pass i
        if (lastPipe >= 0 && lastPipe != k) HIPCHK(c, hipStreamWaitEvent(st, c->pipeFilmDone[lastPipe], 0));
        hipLaunchKernelGGL(k_film_path, dim3(grid), dim3(RT_BLOCK), 0, st, cam, ps, pb, rgb, wsum, rs.spillRgb, rs.spillW);
        HIPCHK(c, hipEventRecord(c->pipeFilmDone[k], st));
        lastPipe = k;
        HIPCHK(c, hipGetLastError());
    }
    // join: the context's stream continues after all pipelines
    for (int k = 0; k < rt_context::kPipes; ++k) {
        HIPCHK(c, hipEventRecord(c->mainReady, c->pipeStream[k]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->mainReady, 0));
    }
    hipLaunchKernelGGL(k_merge_spill, dim3(grid), dim3(RT_BLOCK), 0, c->stream, (size_t)cam.width * cam.height, rgb, wsum, rs.spillRgb, rs.spillW);
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

// Path integrator on the reference's per-tile random stream: one wave (lane 0) per 16x16 tile, see k_path_tile_stream.
int render_path_tile_stream(rt_context *c, const rt_render_params *p, float *rgb, float *wsum) {
    const CameraDev &cam = c->cam;
    const int sbW = cam.sb[2] - cam.sb[0], sbH = cam.sb[3] - cam.sb[1];
    if (sbW <= 0 || sbH <= 0 || p->spp_total == 0) return RT_OK;
    int rc;
    if ((rc = ensure_spill(c))) return rc;
    const uint32_t ntx = (uint32_t)(sbW + 15) / 16, nty = (uint32_t)(sbH + 15) / 16;
    const uint32_t nTiles = ntx * nty;
    const uint32_t wavesPerBlock = RT_BLOCK / 64;
    auto k = RT_PICK(k_path_tile_stream, c, p->collect_stats != 0);
    hipLaunchKernelGGL(k, dim3((nTiles + wavesPerBlock - 1) / wavesPerBlock), dim3(RT_BLOCK), trace_lds_bytes(c), c->stream, c->sc, cam, p->spp_total,
                       p->max_depth, p->rr_threshold, nTiles, ntx, rgb, wsum, c->dSpillRgb, c->dSpillW, c->dStats);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_merge_spill, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, (size_t)cam.width * cam.height, rgb, wsum, c->dSpillRgb,
                       c->dSpillW);
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

int render_whitted(rt_context *c, const rt_render_params *p, float *rgb, float *wsum, const RenderScratch *fixed) {
    const CameraDev &cam = c->cam;
    const int sbW = cam.sb[2] - cam.sb[0], sbH = cam.sb[3] - cam.sb[1];
    if (sbW <= 0 || sbH <= 0 || p->spp_end == p->spp_begin) return RT_OK;
    int rc;
    if (fixed && p->rng_mode == RT_RNG_REFERENCE_TILE_STREAM)
        return fail(c, RT_ERR_UNSUPPORTED, "the reference tile stream needs per-tile buffers: not available without allocation");
    if (!fixed && (rc = ensure_spill(c))) return rc;
    float *spillRgb = fixed ? fixed->spillRgb : c->dSpillRgb, *spillW = fixed ? fixed->spillW : c->dSpillW;
    unsigned long long *statsPtr = fixed ? fixed->stats : c->dStats;
    const bool stats = p->collect_stats != 0;
    const size_t lds = trace_lds_bytes(c);
    const uint32_t ntx = (uint32_t)(sbW + 15) / 16, nty = (uint32_t)(sbH + 15) / 16;
    const uint32_t nTiles = ntx * nty;
    if (p->rng_mode == RT_RNG_REFERENCE_TILE_STREAM) {
        const size_t need = (size_t)sbW * sbH;
        if (c->pixelStateCap < need) {
            if (c->dPixelState) (void)hipFree(c->dPixelState);
            c->dPixelState = nullptr;
            c->pixelStateCap = 0;
            HIPCHK(c, hipMalloc((void **)&c->dPixelState, need * sizeof(uint2)));
            c->pixelStateCap = need;
        }
        // parallel fixed point over per-pixel stream offsets (k_whitted_count / k_whitted_scan), serial kernel as fallback
        const size_t nSlotsPre = (size_t)nTiles * 256;
        if (c->whittedScanCap < nSlotsPre) {
            if (c->dWhittedOffsets) (void)hipFree(c->dWhittedOffsets);
            if (c->dWhittedDraws) (void)hipFree(c->dWhittedDraws);
            c->dWhittedOffsets = c->dWhittedDraws = nullptr;
            c->whittedScanCap = 0;
            HIPCHK(c, hipMalloc((void **)&c->dWhittedOffsets, nSlotsPre * sizeof(unsigned long long) + 256 + (size_t)nTiles * sizeof(uint32_t)));
            HIPCHK(c, hipMalloc((void **)&c->dWhittedDraws, nSlotsPre * sizeof(unsigned long long)));
            c->whittedScanCap = nSlotsPre;
        }
        uint32_t *dChanged = reinterpret_cast<uint32_t *>(c->dWhittedOffsets + nSlotsPre);
        uint32_t *dTileLive = dChanged + 64;
        const int scanGrid = (int)((nTiles + RT_BLOCK - 1) / RT_BLOCK);
        const int countGrid = (int)std::min<size_t>((nSlotsPre + RT_BLOCK - 1) / RT_BLOCK, (size_t)c->gridBlocks * 4u);
        // first guess: every sample hits a surface without specular recursion: 2 camera + 2 per light + 2 + 2 (specularReflect / Transmit)
        const unsigned long long guess = (unsigned long long)p->spp_total * (6ull + 2ull * c->sc.nLights);
        hipLaunchKernelGGL(k_whitted_scan, dim3(scanGrid), dim3(RT_BLOCK), 0, c->stream, cam, nTiles, ntx, (const unsigned long long *)c->dWhittedDraws,
                           c->dWhittedOffsets, guess, dTileLive, dChanged);
        auto kc = c->hasSpheres ? (c->ldsScene ? k_whitted_count<true, true> : k_whitted_count<false, true>)
                                : (c->ldsScene ? k_whitted_count<true, false> : k_whitted_count<false, false>);
        bool converged = false;
        for (int it = 0; it < 64 && !converged; ++it) {
            HIPCHK(c, hipMemsetAsync(dChanged, 0, sizeof(uint32_t), c->stream));
            hipLaunchKernelGGL(kc, dim3(countGrid), dim3(RT_BLOCK), lds, c->stream, c->sc, cam, p->spp_total, p->max_depth, nTiles, ntx,
                               (const unsigned long long *)c->dWhittedOffsets, (const uint32_t *)dTileLive, c->dWhittedDraws, c->dPixelState);
            hipLaunchKernelGGL(k_whitted_scan, dim3(scanGrid), dim3(RT_BLOCK), 0, c->stream, cam, nTiles, ntx,
                               (const unsigned long long *)c->dWhittedDraws, c->dWhittedOffsets, 0ull, dTileLive, dChanged);
            HIPCHK(c, hipGetLastError());
            uint32_t changed = 1;
            HIPCHK(c, hipMemcpyAsync(&changed, dChanged, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            converged = changed == 0;
            c->whittedPasses = it + 1;
        }
        if (!converged) {
            c->whittedPasses = -1;
            auto k = c->hasSpheres ? (c->ldsScene ? k_whitted_prepass<true, true> : k_whitted_prepass<false, true>)
                                   : (c->ldsScene ? k_whitted_prepass<true, false> : k_whitted_prepass<false, false>);
            // one lane per tile: spread the lanes over many small groups of waves
            hipLaunchKernelGGL(k, dim3((nTiles + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), lds, c->stream, c->sc, cam, p->spp_total, p->max_depth,
                               nTiles, ntx, c->dPixelState);
            HIPCHK(c, hipGetLastError());
        }
    }
    const uint32_t nSlots = nTiles * 256u;
    const int grid = (int)std::min<uint32_t>((nSlots + RT_BLOCK - 1) / RT_BLOCK, (uint32_t)c->gridBlocks * 4u);
    auto k = RT_PICK(k_whitted, c, stats);
    hipLaunchKernelGGL(k, dim3(grid), dim3(RT_BLOCK), lds, c->stream, c->sc, cam, p->spp_begin, p->spp_end, p->spp_total, p->max_depth, p->rng_mode,
                       p->seed, (const uint2 *)c->dPixelState, rgb, wsum, spillRgb, spillW, statsPtr);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_merge_spill, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, (size_t)cam.width * cam.height, rgb, wsum, spillRgb,
                       spillW);
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

int render_dispatch(rt_context *c, const rt_render_params *p, float *rgb, float *wsum, const RenderScratch *fixed = nullptr) {
    int rc;
    if (p->integrator == RT_INTEGRATOR_WHITTED) rc = render_whitted(c, p, rgb, wsum, fixed);
    else if (p->rng_mode == RT_RNG_REFERENCE_TILE_STREAM)
        rc = fixed ? fail(c, RT_ERR_UNSUPPORTED, "the reference tile stream is not available without allocation") : render_path_tile_stream(c, p, rgb, wsum);
    else rc = render_path(c, p, rgb, wsum, fixed);
    // every renderer ends with k_merge_spill, which leaves the spill buffers zeroed; after an error return in between
    // they may hold samples of the failed render, and the next render clears them first (ensure_spill)
    if (rc != RT_OK && !fixed) c->spillDirty = true;
    return rc;
}

// PathBuffers of one pipeline laid over the queue part of the scratch array: every per-path array holds `capacity` entries.
PathBuffers carve_path_buffers(unsigned char *base, size_t capacity, uint32_t *counts) {
    PathBuffers pb{};
    float4 *q = reinterpret_cast<float4 *>(base);
    auto take = [&](size_t vecPerPath) { float4 *r = q; q += capacity * vecPerPath; return r; };
    for (int k = 0; k < 2; ++k) { pb.rayA[k] = take(1); pb.rayB[k] = take(1); }
    pb.hit = take(1);
    pb.L = take(1);
    pb.beta = take(1);
    pb.rng = reinterpret_cast<uint4 *>(take(1));
    pb.shA = take(1);
    pb.shB = take(1);
    pb.shC = take(1);
    pb.probe = take(6);
    pb.counts = counts;
    return pb;
}


// ---------------------------------------------------------------------------------------------------
// multi-GPU group: samples-per-pixel split across the members, one RCCL sum-reduce of the packed float film
// ---------------------------------------------------------------------------------------------------
// RCCL is opened at run time, and only by rt_create_multi with more than one distinct device: a single-GPU process
// never loads it. A copy already in the process (a host that brought its own) is preferred over the system's.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rcclMutex;

bool load_rccl(std::string &why) {
    std::lock_guard<std::mutex> lock(g_rcclMutex);
    if (g_rccl.handle) return true;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"})
        if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) { why = std::string("cannot open librccl: ") + dlerror(); return false; }
    RcclApi a;
    a.handle = h;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.Reduce = reinterpret_cast<decltype(a.Reduce)>(dlsym(h, "ncclReduce"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.CommInitAll || !a.CommDestroy || !a.Reduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
        why = "librccl lacks a required entry point";
        return false;
    }
    g_rccl = a;
    return true;
}

#define NCCLCHK(ctx, expr)                                                                        \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r_);                   \
            return RT_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

int ensure_group_film(rt_context *m, size_t nPix) {
    if (m->groupFilmPixels == nPix && m->dGroupFilm) return RT_OK;
    if (m->dGroupFilm) (void)hipFree(m->dGroupFilm);
    m->dGroupFilm = nullptr;
    m->groupFilmPixels = 0;
    HIPCHK(m, hipMalloc((void **)&m->dGroupFilm, nPix * 4 * sizeof(float)));
    m->groupFilmPixels = nPix;
    return RT_OK;
}

// dst += src over n floats
__global__ __launch_bounds__(RT_BLOCK) void k_add_floats(size_t n, float *__restrict__ dst, const float *__restrict__ src) {
    for (size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * RT_BLOCK) dst[i] += src[i];
}

// Render of a group. Member k of G renders sample indices [b + k (e - b) / G, b + (k + 1) (e - b) / G) of every pixel into
// its own zeroed packed film on its own GPU; the films are summed into member 0's with ONE ncclReduce over 4 W H floats
// (members on distinct GPUs) and added to the caller's buffers on device 0. The reference's per-tile random stream is
// serial inside a tile and cannot be cut by sample index: such jobs run on member 0 alone.
int render_group(rt_context *c, const rt_render_params *p, float *rgb, float *wsum) {
    const int G = 1 + (int)c->peers.size();
    const int total = p->spp_end - p->spp_begin;
    if (p->rng_mode == RT_RNG_REFERENCE_TILE_STREAM || total < G) return render_dispatch(c, p, rgb, wsum);
    const size_t nPix = (size_t)c->cam.width * c->cam.height;
    std::vector<rt_context *> members{c};
    members.insert(members.end(), c->peers.begin(), c->peers.end());
    int rc;
    for (int k = 0; k < G; ++k) {
        rt_context *m = members[k];
        HIPCHK(c, hipSetDevice(m->device));
        if ((rc = ensure_group_film(m, nPix))) { c->err = m->err; return rc; }
        HIPCHK(c, hipMemsetAsync(m->dGroupFilm, 0, nPix * 4 * sizeof(float), m->stream));
        rt_render_params pk = *p;
        rt_shard_range(p->spp_begin, p->spp_end, k, G, &pk.spp_begin, &pk.spp_end);
        if ((rc = render_dispatch(m, &pk, m->dGroupFilm, m->dGroupFilm + 3 * nPix))) { c->err = m->err; return rc; }
    }
    if (!c->comms.empty()) {
        // distinct GPUs: every member's stream continues with its part of the collective, after its own render
        NCCLCHK(c, g_rccl.GroupStart());
        for (int k = 0; k < G; ++k) {
            rt_context *m = members[k];
            ncclResult_t r = hipSetDevice(m->device) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
            if (r == ncclSuccess) r = g_rccl.Reduce(m->dGroupFilm, m->dGroupFilm, nPix * 4, ncclFloat, ncclSum, 0, c->comms[k], m->stream);
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();   // a group that was started is ended, whatever happened inside it
                (void)hipSetDevice(c->device);
                c->err = std::string("ncclReduce of member ") + std::to_string(k) + ": " + g_rccl.GetErrorString(r);
                return RT_ERR_HIP;
            }
        }
        NCCLCHK(c, g_rccl.GroupEnd());
        HIPCHK(c, hipSetDevice(c->device));
    } else {
        // rehearsal on one GPU (the same device listed more than once; RCCL refuses that): device-local sums in member order
        HIPCHK(c, hipSetDevice(c->device));
        for (int k = 1; k < G; ++k) {
            rt_context *m = members[k];
            HIPCHK(c, hipEventRecord(m->groupDone, m->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream, m->groupDone, 0));
            hipLaunchKernelGGL(k_add_floats, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, nPix * 4, c->dGroupFilm, (const float *)m->dGroupFilm);
        }
        // the sums above read the peers' films on the root's stream: a peer's stream (whose next render clears its film)
        // continues only after them
        HIPCHK(c, hipEventRecord(c->groupDone, c->stream));
        for (int k = 1; k < G; ++k) HIPCHK(c, hipStreamWaitEvent(members[k]->stream, c->groupDone, 0));
    }
    hipLaunchKernelGGL(k_add_floats, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, nPix * 3, rgb, (const float *)c->dGroupFilm);
    hipLaunchKernelGGL(k_add_floats, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, nPix, wsum, (const float *)(c->dGroupFilm + 3 * nPix));
    HIPCHK(c, hipGetLastError());
    // the peers' streams are drained before the call returns: their next render reuses the packed films
    for (int k = 1; k < G; ++k) {
        HIPCHK(c, hipSetDevice(members[k]->device));
        HIPCHK(c, hipStreamSynchronize(members[k]->stream));
    }
    HIPCHK(c, hipSetDevice(c->device));
    return RT_OK;
}

int render_any(rt_context *c, const rt_render_params *p, float *rgb, float *wsum) {
    return c->peers.empty() ? render_dispatch(c, p, rgb, wsum) : render_group(c, p, rgb, wsum);
}

}  // namespace

extern "C" {

const char *rt_version(void) { return "raytracing_amd 0.1 (gfx950)"; }

const char *rt_last_error(const rt_context *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int rt_create(int device_id, rt_context **out) {
    if (!out) { g_create_err = "out_ctx is NULL"; return RT_ERR_INVALID_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_err = std::string("no HIP device is visible (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                       "); this library has no CPU fallback";
        return RT_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) { g_create_err = "device_id out of range"; return RT_ERR_INVALID_ARG; }
    auto *c = new rt_context();
    c->device = device_id;
    auto bail = [&](const char *what, hipError_t err) {
        g_create_err = std::string(what) + ": " + hipGetErrorString(err);
        rt_destroy(c);   // frees whatever has been created so far
        return RT_ERR_HIP;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#ifndef RT_GRID_BLOCKS_PER_CU
#define RT_GRID_BLOCKS_PER_CU 8   // persistent grids: workgroups per CU (grid-stride loops over the device-resident queue counters)
#endif
    c->gridBlocks = c->numCUs * RT_GRID_BLOCKS_PER_CU;
    if ((e = scratch_address(device_id, &c->scratchPtr)) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (int k = 0; k < rt_context::kPipes; ++k) {
        if ((e = hipStreamCreateWithFlags(&c->pipeStream[k], hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
        if ((e = hipEventCreateWithFlags(&c->pipeFilmDone[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    if ((e = hipEventCreateWithFlags(&c->mainReady, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc((void **)&c->dStats, ST_COUNT * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemsetAsync(c->dStats, 0, ST_COUNT * sizeof(unsigned long long), c->stream)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
    *out = c;
    return RT_OK;
}

static void destroy_one(rt_context *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < rt_context::kPipes; ++k) {
        if (c->pipeStream[k]) { (void)hipStreamSynchronize(c->pipeStream[k]); (void)hipStreamDestroy(c->pipeStream[k]); }
        if (c->pipeFilmDone[k]) (void)hipEventDestroy(c->pipeFilmDone[k]);
    }
    if (c->mainReady) (void)hipEventDestroy(c->mainReady);
    free_scene(c);
    free_path_buffers(c);
    if (c->dStats) (void)hipFree(c->dStats);
    if (c->dSpillRgb) (void)hipFree(c->dSpillRgb);
    if (c->dSpillW) (void)hipFree(c->dSpillW);
    if (c->dPixelState) (void)hipFree(c->dPixelState);
    if (c->dWhittedOffsets) (void)hipFree(c->dWhittedOffsets);
    if (c->dWhittedDraws) (void)hipFree(c->dWhittedDraws);
    if (c->dGroupFilm) (void)hipFree(c->dGroupFilm);
    if (c->groupDone) (void)hipEventDestroy(c->groupDone);
    for (auto *events : {&c->extendEvents, &c->shadeEvents})
        for (auto &ep : *events) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int upload_scene_one(rt_context *c, const rt_scene_desc *d) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!d || !d->nodes || !d->tri_p || !d->material_id || !d->light_id || !d->materials || d->n_nodes == 0 || d->n_prims == 0 ||
        d->n_materials == 0 || (d->n_leaf_idx && !d->leaf_idx) || (d->n_lights && !d->lights))
        return fail(c, RT_ERR_INVALID_ARG, "scene description has NULL or empty required arrays");
    // validate everything a kernel will index with
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        if (d->shape_type && d->shape_type[i] != RT_SHAPE_TRIANGLE && d->shape_type[i] != RT_SHAPE_SPHERE)
            return fail(c, RT_ERR_INVALID_ARG, "unknown shape type");
        if (d->material_id[i] < 0 || (uint32_t)d->material_id[i] >= d->n_materials) return fail(c, RT_ERR_INVALID_ARG, "material_id out of range");
        if (d->light_id[i] >= (int32_t)d->n_lights) return fail(c, RT_ERR_INVALID_ARG, "light_id out of range");
    }
    // every sphere primitive needs exactly one rt_sphere record
    std::vector<int32_t> sphereOf(d->n_prims, -1);
    if (d->n_spheres && !d->spheres) return fail(c, RT_ERR_INVALID_ARG, "spheres is NULL");
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        const int32_t pid = d->spheres[i].prim_id;
        if (pid < 0 || (uint32_t)pid >= d->n_prims || !d->shape_type || d->shape_type[pid] != RT_SHAPE_SPHERE || sphereOf[pid] >= 0)
            return fail(c, RT_ERR_INVALID_ARG, "sphere record does not match a sphere primitive");
        if (!(d->spheres[i].radius > 0)) return fail(c, RT_ERR_INVALID_ARG, "sphere radius must be positive");
        sphereOf[pid] = (int32_t)i;
    }
    for (uint32_t i = 0; i < d->n_prims; ++i)
        if (d->shape_type && d->shape_type[i] == RT_SHAPE_SPHERE && sphereOf[i] < 0)
            return fail(c, RT_ERR_INVALID_ARG, "sphere primitive without an rt_sphere record");
    for (uint32_t i = 0; i < d->n_lights; ++i)
        if (d->lights[i].prim_id < 0 || (uint32_t)d->lights[i].prim_id >= d->n_prims) return fail(c, RT_ERR_INVALID_ARG, "light prim_id out of range");
    for (uint32_t i = 0; i < d->n_materials; ++i)
        if (d->materials[i].type < RT_MAT_LAMBERTIAN || d->materials[i].type > RT_MAT_GLASS) return fail(c, RT_ERR_INVALID_ARG, "unknown material type");
    // Depth of every node below the root (children have larger indices than their parent, so one forward pass fills it;
    // a node that several parents name, which no builder produces but the format can express, counts with its deepest
    // parent). An interior node at depth k can find k entries pending on the todo stack and push one more: the traversal
    // keeps RT_STACK_TOTAL entries, like the reference's todo[64] (Accelerators/KDTree.cpp:396-398).
    std::vector<uint8_t> depth(d->n_nodes, 0);
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
        const uint32_t w0 = d->nodes[2 * i], w1 = d->nodes[2 * i + 1];
        if ((w1 & 3u) == 3u) {
            const uint32_t n = w1 >> 2;
            if (n == 1) { if (w0 >= d->n_prims) return fail(c, RT_ERR_INVALID_ARG, "leaf primitive index out of range"); }
            else if (n > 1 && ((uint64_t)w0 + n > d->n_leaf_idx)) return fail(c, RT_ERR_INVALID_ARG, "leaf index range out of bounds");
        } else {
            const uint32_t above = w1 >> 2;
            if (i + 1 >= d->n_nodes || above >= d->n_nodes || above <= i) return fail(c, RT_ERR_INVALID_ARG, "interior node child out of range");
            if (depth[i] >= RT_STACK_TOTAL) return fail(c, RT_ERR_INVALID_ARG, "KD-tree deeper than the 64-entry todo stack");
            const uint8_t below = (uint8_t)(depth[i] + 1);
            depth[i + 1] = std::max(depth[i + 1], below);
            depth[above] = std::max(depth[above], below);
        }
    }
    for (uint32_t i = 0; i < d->n_leaf_idx; ++i)
        if (d->leaf_idx[i] < 0 || (uint32_t)d->leaf_idx[i] >= d->n_prims) return fail(c, RT_ERR_INVALID_ARG, "leaf_idx entry out of range");

    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_scene(c);

    const uint32_t np = d->n_prims;
    // scenes whose traversal tables fit in LDS carry the triangle positions in all three axis permutations
    const size_t ldsBytes = (((size_t)d->n_nodes * 8 + 15) / 16 + ((size_t)d->n_leaf_idx * 4 + 15) / 16 + (size_t)np * 3 * 3) * 16;
    c->ldsScene = ldsBytes <= 40 * 1024;
    const uint32_t triCopies = c->ldsScene ? 3u : 1u;
    std::vector<float4> hot((size_t)np * 3 * triCopies), shade((size_t)np * 5);
    for (uint32_t i = 0; i < np; ++i) {
        if (sphereOf[i] >= 0) {
            float bits, mi, li;
            std::memcpy(&bits, &sphereOf[i], 4);
            for (uint32_t b = 0; b < triCopies; ++b) {
                hot[((size_t)b * np + i) * 3 + 0] = make_float4(bits, bits, bits, RT_PRIM_SPHERE);
                hot[((size_t)b * np + i) * 3 + 1] = hot[((size_t)b * np + i) * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int k = 0; k < 4; ++k) shade[5 * (size_t)i + k] = make_float4(0.f, 0.f, 0.f, 0.f);
            std::memcpy(&mi, &d->material_id[i], 4);
            std::memcpy(&li, &d->light_id[i], 4);
            shade[5 * (size_t)i + 4] = make_float4(mi, li, 0.f, 0.f);
            continue;
        }
        const float *p = &d->tri_p[9 * (size_t)i];
        f3 ss, ng;
        bool degenerate;
        tri_constants(p, d->tri_uv ? &d->tri_uv[6 * (size_t)i] : nullptr, ss, ng, degenerate);
        hot[3 * (size_t)i + 0] = make_float4(p[0], p[1], p[2], degenerate ? 1.f : 0.f);
        hot[3 * (size_t)i + 1] = make_float4(p[3], p[4], p[5], 0.f);
        hot[3 * (size_t)i + 2] = make_float4(p[6], p[7], p[8], 0.f);
        for (uint32_t b = 1; b < triCopies; ++b) {   // block b serves kz = b - 1: (y,z,x) for kz 0, (z,x,y) for kz 1
            const int kx = (int)b % 3, ky = (kx + 1) % 3, kzz = (int)b - 1;
            for (int v = 0; v < 3; ++v)
                hot[((size_t)b * np + i) * 3 + v] = make_float4(p[3 * v + kx], p[3 * v + ky], p[3 * v + kzz], v == 0 && degenerate ? 1.f : 0.f);
        }
        const bool hasN = d->tri_n && (!d->tri_has_n || d->tri_has_n[i]);
        const float *nn = hasN ? &d->tri_n[9 * (size_t)i] : nullptr;
        shade[5 * (size_t)i + 0] = make_float4(nn ? nn[0] : 0.f, nn ? nn[1] : 0.f, nn ? nn[2] : 0.f, ss.x);
        shade[5 * (size_t)i + 1] = make_float4(nn ? nn[3] : 0.f, nn ? nn[4] : 0.f, nn ? nn[5] : 0.f, ss.y);
        shade[5 * (size_t)i + 2] = make_float4(nn ? nn[6] : 0.f, nn ? nn[7] : 0.f, nn ? nn[8] : 0.f, ss.z);
        shade[5 * (size_t)i + 3] = make_float4(ng.x, ng.y, ng.z, hasN ? 1.f : 0.f);
        float mi, li;
        std::memcpy(&mi, &d->material_id[i], 4);
        std::memcpy(&li, &d->light_id[i], 4);
        shade[5 * (size_t)i + 4] = make_float4(mi, li, 0.f, 0.f);
    }
    std::vector<DevMaterial> mats(d->n_materials);
    c->hasGlass = false;
    c->allLambert = true;
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->materials[i].type != RT_MAT_LAMBERTIAN) c->allLambert = false;
        mats[i] = DevMaterial{d->materials[i].rgb[0], d->materials[i].rgb[1], d->materials[i].rgb[2], d->materials[i].eta, d->materials[i].type, 0, 0, 0};
        if (d->materials[i].type == RT_MAT_GLASS) c->hasGlass = true;
    }
    std::vector<DevLight> lights(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const rt_light &l = d->lights[i];
        if (sphereOf[l.prim_id] >= 0) {
            const float r = d->spheres[sphereOf[l.prim_id]].radius;
            lights[i] = DevLight{l.emit[0], l.emit[1], l.emit[2], l.two_sided, l.prim_id, 4.0f * RT_PI * r * r, 0.f, 0.f};   // ASphereShape::area
            continue;
        }
        const float *p = &d->tri_p[9 * (size_t)l.prim_id];
        const f3 p0 = mk3(p[0], p[1], p[2]), p1 = mk3(p[3], p[4], p[5]), p2 = mk3(p[6], p[7], p[8]);
        const f3 cr = cross3(p1 - p0, p2 - p0);
        const float area = (float)(0.5 * std::sqrt(dot3(cr, cr)));   // ATriangleShape::area (TriangleShape.cpp:153-160)
        lights[i] = DevLight{l.emit[0], l.emit[1], l.emit[2], l.two_sided, l.prim_id, area, 0.f, 0.f};
    }
    // UniformLightDistribution / Distribution1D (Utils/LightDistrib.cpp:17-21, LightDistrib.h:14-33)
    const int nl = (int)d->n_lights;
    std::vector<float> func(std::max(nl, 1), 1.0f), cdf(nl + 1, 0.f);
    float funcInt = 0;
    if (nl > 0) {
        for (int i = 1; i < nl + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / nl;
        funcInt = cdf[nl];
        if (funcInt == 0) for (int i = 1; i < nl + 1; ++i) cdf[i] = float(i) / float(nl);
        else for (int i = 1; i < nl + 1; ++i) cdf[i] /= funcInt;
    }

    std::vector<uint32_t> nodes(d->nodes, d->nodes + 2 * (size_t)d->n_nodes);
    std::vector<int32_t> leafIdx;
    if (d->n_leaf_idx) leafIdx.assign(d->leaf_idx, d->leaf_idx + d->n_leaf_idx);
    int rc;
    if ((rc = upload(c, &c->dNodes, nodes))) return rc;
    if ((rc = upload(c, &c->dLeafIdx, leafIdx))) return rc;
    if ((rc = upload(c, &c->dTriHot, hot))) return rc;
    if ((rc = upload(c, &c->dTriShade, shade))) return rc;
    if ((rc = upload(c, &c->dMats, mats))) return rc;
    if ((rc = upload(c, &c->dLights, lights))) return rc;
    if ((rc = upload(c, &c->dLightCdf, cdf))) return rc;
    if ((rc = upload(c, &c->dLightFunc, func))) return rc;
    std::vector<DevSphere> spheres(d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        std::memcpy(spheres[i].w2o, d->spheres[i].world_to_object, 64);
        std::memcpy(spheres[i].o2w, d->spheres[i].object_to_world, 64);
        spheres[i].radius = d->spheres[i].radius;
        spheres[i].pad0 = spheres[i].pad1 = spheres[i].pad2 = 0.f;
    }
    if ((rc = upload(c, &c->dSpheres, spheres))) return rc;

    DevScene &sc = c->sc;
    sc.nodes = (const uint2 *)c->dNodes;
    sc.leafIdx = (const int *)c->dLeafIdx;
    sc.triHot = (const float4 *)c->dTriHot;
    sc.triShade = (const float4 *)c->dTriShade;
    sc.mats = (const DevMaterial *)c->dMats;
    sc.lights = (const DevLight *)c->dLights;
    sc.lightCdf = (const float *)c->dLightCdf;
    sc.lightFunc = (const float *)c->dLightFunc;
    sc.spheres = (const DevSphere *)c->dSpheres;
    c->hasSpheres = d->n_spheres > 0;
    sc.nSpheres = d->n_spheres;
    // shading tables in LDS when they are small (cornellBox: 5 KB); beyond 16 KB the shade kernel reads them from memory
    {
        const size_t bytes = 16 * (size_t)shade_tables_vecs(d->n_prims, d->n_materials, d->n_lights, d->n_spheres);
        c->shadeLdsBytes = (d->n_prims < (1u << 16) && bytes <= 16384) ? bytes : 0;
    }
    for (int k = 0; k < 3; ++k) { sc.bmin[k] = d->bounds[k]; sc.bmax[k] = d->bounds[3 + k]; }
    sc.nNodes = d->n_nodes;
    sc.nLeafIdx = d->n_leaf_idx;
    sc.nPrims = np;
    sc.nMats = d->n_materials;
    sc.nLights = d->n_lights;
    sc.lightFuncInt = funcInt;
    sc.triCopies = triCopies;
    c->residentGrid.clear();
    sc.ldsSceneBytes = c->ldsScene ? (uint32_t)ldsBytes : 0;
    c->haveScene = true;
    // resident grids of the streaming render stages, looked up now so that a render does not have to (the lookup fills a
    // host-side table; rt_render_noalloc_dev allocates nothing, on the host side either)
    (void)resident_grid(c, RT_PICK(k_extend_stream, c, false), stream_lds_bytes_of(c));
    (void)resident_grid(c, RT_PICK(k_shadow_stream, c, false), stream_lds_bytes_of(c));
    (void)resident_grid(c, RT_PICK(k_extend_refill, c, false), trace_lds_bytes(c));
    (void)resident_grid(c, RT_PICK(k_extend_refill, c, true), trace_lds_bytes(c));
    (void)resident_grid(c, RT_PICK(k_shadow_refill, c, false), trace_lds_bytes(c));
    (void)resident_grid(c, RT_PICK(k_shadow_refill, c, true), trace_lds_bytes(c));
    return RT_OK;
}

static int set_camera_one(rt_context *c, const float r2c[16], const float c2w[16], int width, int height, const int sb[4]) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!r2c || !c2w || !sb || width <= 0 || height <= 0) return fail(c, RT_ERR_INVALID_ARG, "bad camera argument");
    // Film::getSampleBounds reaches past the film by ceil(radius - 0.5) pixels (Object/Film.cpp:79-85)
    constexpr int kMaxOutside = 64;
    if (sb[0] < -kMaxOutside || sb[1] < -kMaxOutside || sb[2] > width + kMaxOutside || sb[3] > height + kMaxOutside || sb[0] > sb[2] ||
        sb[1] > sb[3])
        return fail(c, RT_ERR_INVALID_ARG, "sample bounds must be ordered and lie within 64 pixels of the film");
    std::memcpy(c->cam.r2c, r2c, 64);
    std::memcpy(c->cam.c2w, c2w, 64);
    c->cam.width = width;
    c->cam.height = height;
    std::memcpy(c->cam.sb, sb, 16);
    // film defaults: 0.5 box filter (sample bounds = cropped pixel bounds), no luminance clamp
    c->cam.filterRadius[0] = c->cam.filterRadius[1] = 0.5f;
    c->cam.cb[0] = std::max(sb[0], 0);
    c->cam.cb[1] = std::max(sb[1], 0);
    c->cam.cb[2] = std::min(sb[2], width);
    c->cam.cb[3] = std::min(sb[3], height);
    c->cam.maxSampleLuminance = std::numeric_limits<float>::infinity();
    c->haveCamera = true;
    return RT_OK;
}

static int set_film_one(rt_context *c, float radius_x, float radius_y, const int cb[4], float max_sample_luminance) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->haveCamera) return fail(c, RT_ERR_NO_CAMERA, "rt_set_film follows rt_set_camera");
    if (!cb || !(radius_x > 0) || !(radius_y > 0) || radius_x > 32.f || radius_y > 32.f) return fail(c, RT_ERR_INVALID_ARG, "bad film argument");
    if (cb[0] < 0 || cb[1] < 0 || cb[2] > c->cam.width || cb[3] > c->cam.height || cb[0] > cb[2] || cb[1] > cb[3])
        return fail(c, RT_ERR_INVALID_ARG, "cropped pixel bounds must lie inside the film");
    if (std::isnan(max_sample_luminance)) return fail(c, RT_ERR_INVALID_ARG, "bad film argument");
    c->cam.filterRadius[0] = radius_x;
    c->cam.filterRadius[1] = radius_y;
    std::memcpy(c->cam.cb, cb, 16);
    c->cam.maxSampleLuminance = max_sample_luminance;
    return RT_OK;
}

int rt_trace_closest_dev(rt_context *c, uint64_t n, const float *o, const float *d, const float *tmax, int32_t *prim, float *t, float *bary) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    if (n && (!o || !d)) return fail(c, RT_ERR_INVALID_ARG, "ray arrays are NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return trace_closest_dev(c, n, o, d, tmax, prim, t, bary, false);
}

int rt_trace_any_dev(rt_context *c, uint64_t n, const float *o, const float *d, const float *tmax, uint8_t *occ) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    if (n && (!o || !d || !occ)) return fail(c, RT_ERR_INVALID_ARG, "ray or output arrays are NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return trace_any_dev(c, n, o, d, tmax, occ, false);
}

int rt_trace_closest(rt_context *c, uint64_t n, const float *o, const float *d, const float *tmax, int32_t *prim, float *t, float *bary) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return RT_OK;
    if (!o || !d) return fail(c, RT_ERR_INVALID_ARG, "ray arrays are NULL");
    HIPCHK(c, hipSetDevice(c->device));
    float *dO = nullptr, *dD = nullptr, *dT = nullptr, *dt = nullptr, *dB = nullptr;
    int32_t *dP = nullptr;
    int rc = RT_OK;
    auto cleanup = [&]() { for (void *p : {(void *)dO, (void *)dD, (void *)dT, (void *)dt, (void *)dB, (void *)dP}) if (p) (void)hipFree(p); };
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return RT_ERR_HIP; } } while (0)
    CK(hipMalloc((void **)&dO, n * 12));
    CK(hipMalloc((void **)&dD, n * 12));
    CK(hipMemcpy(dO, o, n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(dD, d, n * 12, hipMemcpyHostToDevice));
    if (tmax) { CK(hipMalloc((void **)&dT, n * 4)); CK(hipMemcpy(dT, tmax, n * 4, hipMemcpyHostToDevice)); }
    if (prim) CK(hipMalloc((void **)&dP, n * 4));
    if (t) CK(hipMalloc((void **)&dt, n * 4));
    if (bary) CK(hipMalloc((void **)&dB, n * 12));
    rc = trace_closest_dev(c, n, dO, dD, dT, dP, dt, dB, true);
    if (rc == RT_OK) {
        CK(hipStreamSynchronize(c->stream));
        if (prim) CK(hipMemcpy(prim, dP, n * 4, hipMemcpyDeviceToHost));
        if (t) CK(hipMemcpy(t, dt, n * 4, hipMemcpyDeviceToHost));
        if (bary) CK(hipMemcpy(bary, dB, n * 12, hipMemcpyDeviceToHost));
    }
    cleanup();
    return rc;
}

int rt_trace_any(rt_context *c, uint64_t n, const float *o, const float *d, const float *tmax, uint8_t *occ) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return RT_OK;
    if (!o || !d || !occ) return fail(c, RT_ERR_INVALID_ARG, "ray or output arrays are NULL");
    HIPCHK(c, hipSetDevice(c->device));
    float *dO = nullptr, *dD = nullptr, *dT = nullptr;
    uint8_t *dOcc = nullptr;
    int rc = RT_OK;
    auto cleanup = [&]() { for (void *p : {(void *)dO, (void *)dD, (void *)dT, (void *)dOcc}) if (p) (void)hipFree(p); };
    CK(hipMalloc((void **)&dO, n * 12));
    CK(hipMalloc((void **)&dD, n * 12));
    CK(hipMemcpy(dO, o, n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(dD, d, n * 12, hipMemcpyHostToDevice));
    if (tmax) { CK(hipMalloc((void **)&dT, n * 4)); CK(hipMemcpy(dT, tmax, n * 4, hipMemcpyHostToDevice)); }
    CK(hipMalloc((void **)&dOcc, n));
    rc = trace_any_dev(c, n, dO, dD, dT, dOcc, true);
    if (rc == RT_OK) {
        CK(hipStreamSynchronize(c->stream));
        CK(hipMemcpy(occ, dOcc, n, hipMemcpyDeviceToHost));
    }
    cleanup();
    return rc;
}

int rt_render_dev(rt_context *c, const rt_render_params *p, float *rgb, float *wsum) {
    int rc = check_render_args(c, p);
    if (rc) return rc;
    if (!rgb || !wsum) return fail(c, RT_ERR_INVALID_ARG, "output buffers are NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return render_any(c, p, rgb, wsum);
}

int rt_render_noalloc_dev(rt_context *c, const rt_render_params *p, float *rgb, float *wsum) {
    int rc = check_render_args(c, p);
    if (rc) return rc;
    if (!rgb || !wsum) return fail(c, RT_ERR_INVALID_ARG, "output buffers are NULL");
    if (p->rng_mode != RT_RNG_COUNTER) return fail(c, RT_ERR_UNSUPPORTED, "rt_render_noalloc_dev: only RT_RNG_COUNTER (the tile stream needs per-tile buffers)");
    if (p->collect_stats) return fail(c, RT_ERR_UNSUPPORTED, "rt_render_noalloc_dev: statistics are not collected");
    if (p->max_depth > RT_NOALLOC_MAX_DEPTH) return fail(c, RT_ERR_UNSUPPORTED, "rt_render_noalloc_dev: max_depth above RT_NOALLOC_MAX_DEPTH");
    const size_t filmPixels = (size_t)c->cam.width * c->cam.height;
    const size_t samplePixels = (size_t)(c->cam.sb[2] - c->cam.sb[0]) * (size_t)(c->cam.sb[3] - c->cam.sb[1]);
    if (filmPixels > noalloc::kMaxPaths || samplePixels > noalloc::kMaxPaths)
        return fail(c, RT_ERR_UNSUPPORTED, "rt_render_noalloc_dev: film or sample bounds above RT_NOALLOC_MAX_PATHS pixels");
    if (!c->scratchPtr) return fail(c, RT_ERR_HIP, "the scratch array has no address on this device");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned char *ptr = c->scratchPtr;   // the device address from hipGetSymbolAddress, never the array's name
    // what must start at zero: the film spill buffers (samples that round onto a pixel border are staged there) and the
    // statistics counters; the queue counters are cleared per pass by the renderer, the queues are written before read
    HIPCHK(c, hipMemsetAsync(ptr + noalloc::kStatsOff, 0, noalloc::kCountsOff - noalloc::kStatsOff, c->stream));
    HIPCHK(c, hipMemsetAsync(ptr + noalloc::kSpillRgbOff, 0, filmPixels * 3 * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(ptr + noalloc::kSpillWOff, 0, filmPixels * sizeof(float), c->stream));
    PathBuffers pb = carve_path_buffers(ptr + noalloc::kQueuesOff, noalloc::kMaxPaths, reinterpret_cast<uint32_t *>(ptr + noalloc::kCountsOff));
    const RenderScratch rs{&pb, 1, noalloc::kMaxPaths, reinterpret_cast<float *>(ptr + noalloc::kSpillRgbOff),
                           reinterpret_cast<float *>(ptr + noalloc::kSpillWOff), reinterpret_cast<unsigned long long *>(ptr + noalloc::kStatsOff),
                           false};
    return render_dispatch(c, p, rgb, wsum, &rs);
}

int rt_render(rt_context *c, const rt_render_params *p, float *rgb, float *wsum) {
    int rc = check_render_args(c, p);
    if (rc) return rc;
    if (!rgb || !wsum) return fail(c, RT_ERR_INVALID_ARG, "output buffers are NULL");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->cam.width * c->cam.height;
    float *dRgb = nullptr, *dW = nullptr;
    auto cleanup = [&]() { if (dRgb) (void)hipFree(dRgb); if (dW) (void)hipFree(dW); };
    CK(hipMalloc((void **)&dRgb, n * 12));
    CK(hipMalloc((void **)&dW, n * 4));
    CK(hipMemcpy(dRgb, rgb, n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, wsum, n * 4, hipMemcpyHostToDevice));
    const auto t0 = std::chrono::steady_clock::now();
    rc = render_any(c, p, dRgb, dW);
    if (rc == RT_OK) {
        CK(hipStreamSynchronize(c->stream));
        c->renderMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipMemcpy(rgb, dRgb, n * 12, hipMemcpyDeviceToHost));
        CK(hipMemcpy(wsum, dW, n * 4, hipMemcpyDeviceToHost));
    }
    cleanup();
    return rc;
}
#undef CK

static int synchronize_one(rt_context *c) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

void *rt_stream(rt_context *c) { return c ? (void *)c->stream : nullptr; }

int rt_develop_dev(rt_context *c, const float *rgb, const float *wsum, float film_scale, uint8_t *rgb8) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!rgb || !wsum || !rgb8) return fail(c, RT_ERR_INVALID_ARG, "NULL buffer");
    if (!c->haveCamera) return fail(c, RT_ERR_NO_CAMERA, "no camera set (the film size comes from rt_set_camera)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->cam.width * c->cam.height;
    hipLaunchKernelGGL(k_develop, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, n, rgb, wsum, film_scale, rgb8);
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

int rt_develop(rt_context *c, const float *rgb, const float *wsum, float film_scale, uint8_t *rgb8) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!rgb || !wsum || !rgb8) return fail(c, RT_ERR_INVALID_ARG, "NULL buffer");
    if (!c->haveCamera) return fail(c, RT_ERR_NO_CAMERA, "no camera set (the film size comes from rt_set_camera)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->cam.width * c->cam.height;
    float *dR = nullptr, *dW = nullptr;
    uint8_t *d8 = nullptr;
    int rc = RT_OK;
    auto cleanup = [&]() { if (dR) (void)hipFree(dR); if (dW) (void)hipFree(dW); if (d8) (void)hipFree(d8); };
#define DEV_TRY(expr)                                                                       \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(c, RT_ERR_HIP, hipGetErrorString(e_)); cleanup(); return rc; } } while (0)
    DEV_TRY(hipMalloc((void **)&dR, n * 3 * sizeof(float)));
    DEV_TRY(hipMalloc((void **)&dW, n * sizeof(float)));
    DEV_TRY(hipMalloc((void **)&d8, n * 3));
    DEV_TRY(hipMemcpy(dR, rgb, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    DEV_TRY(hipMemcpy(dW, wsum, n * sizeof(float), hipMemcpyHostToDevice));
    rc = rt_develop_dev(c, dR, dW, film_scale, d8);
    if (rc == RT_OK) {
        DEV_TRY(hipStreamSynchronize(c->stream));
        DEV_TRY(hipMemcpy(rgb8, d8, n * 3, hipMemcpyDeviceToHost));
    }
#undef DEV_TRY
    cleanup();
    return rc;
}

static int set_option_one(rt_context *c, int option, int value) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (option == RT_OPTION_TRACE_SCHEDULE) {
        if (value != RT_SCHEDULE_AUTO && value != RT_SCHEDULE_LOCKSTEP && value != RT_SCHEDULE_STREAM && value != RT_SCHEDULE_REFILL)
            return fail(c, RT_ERR_INVALID_ARG, "unknown trace schedule");
        c->traceSchedule = value;
        return RT_OK;
    }
    if (option == RT_OPTION_PIPELINES) {
        if (value < 1 || value > rt_context::kPipes) return fail(c, RT_ERR_INVALID_ARG, "pipelines must be 1..RT_MAX_PIPELINES");
        c->pipelines = value;
        return RT_OK;
    }
    if (option == RT_OPTION_KERNEL_TIMES) {
        if (value != 0 && value != 1) return fail(c, RT_ERR_INVALID_ARG, "kernel times: 0 or 1");
        c->kernelTimes = value != 0;
        return RT_OK;
    }
    return fail(c, RT_ERR_INVALID_ARG, "unknown option");
}

static int get_stats_one(rt_context *c, rt_stats *out) {
    if (!c || !out) return RT_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long h[ST_COUNT];
    HIPCHK(c, hipMemcpy(h, c->dStats, sizeof h, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->rays_closest = h[ST_RAYS_CLOSEST];
    out->rays_any = h[ST_RAYS_ANY];
    out->camera_samples = h[ST_CAMERA_SAMPLES];
    out->n_interior = h[ST_INTERIOR];
    out->n_leaf = h[ST_LEAF];
    out->n_leaf_idx = h[ST_LEAFIDX];
    out->n_tri_tests = h[ST_TRI];
    out->counted_rays = h[ST_COUNTED_RAYS];
    out->extend_rays = h[ST_EXTEND_RAYS];
    const double ms = elapsed_ms(c->extendEvents, c->extendEventsUsed);
    out->extend_launches = c->extendEventsUsed;
    out->extend_ms = ms;
    out->shade_launches = c->shadeEventsUsed;
    out->shade_ms = elapsed_ms(c->shadeEvents, c->shadeEventsUsed);
    out->render_ms = c->renderMs;
    out->queue_bytes_per_pipeline = c->pbBytesPerPipe;
    out->queue_paths = c->pbCapacity;
    out->pipelines = c->pipelines;
    out->scene_in_lds = c->haveScene && c->ldsScene ? 1 : 0;
    return RT_OK;
}

static int reset_stats_one(rt_context *c) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemsetAsync(c->dStats, 0, ST_COUNT * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->extendEventsUsed = c->shadeEventsUsed = 0;
    c->renderMs = 0;
    return RT_OK;
}

// Host evaluation of the device's sin/cos for the concentric disk mapping, for CPU-side accuracy tests.
// Host-side evaluation of the device's trigonometric helpers (they are __host__ __device__): lets a CPU test compare
// them with a double-precision libm without a GPU.
void rt_debug_sincos_disk(float theta, float *s, float *c) { sincos_disk(theta, *s, *c); }
int rt_debug_whitted_passes(const rt_context *c) { return c ? c->whittedPasses : 0; }
void rt_debug_sincos_cr(float theta, float *s, float *c) { sincos_cr(theta, *s, *c); }


// ---- entry points that address every member of a group ------------------------------------------------------------

int rt_create_multi(const int *device_ids, int n_devices, rt_context **out) {
    if (!out) { g_create_err = "out_ctx is NULL"; return RT_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > noalloc::kMaxDevices) { g_create_err = "bad device list"; return RT_ERR_INVALID_ARG; }
    bool distinct = true, same = true;
    for (int k = 0; k < n_devices; ++k) {
        if (device_ids[k] != device_ids[0]) same = false;
        for (int j = 0; j < k; ++j)
            if (device_ids[k] == device_ids[j]) distinct = false;
    }
    // all different (one RCCL rank per GPU) or all the same (the one-GPU rehearsal, films summed on that device); a list that
    // repeats some devices only would have the rehearsal's device-local sums read another GPU's memory
    if (!distinct && !same) { g_create_err = "device list must name distinct devices, or one device repeatedly"; return RT_ERR_INVALID_ARG; }
    rt_context *root = nullptr;
    int rc = rt_create(device_ids[0], &root);
    if (rc) return rc;
    for (int k = 1; k < n_devices; ++k) {
        rt_context *m = nullptr;
        if ((rc = rt_create(device_ids[k], &m))) { rt_destroy(root); return rc; }
        m->groupRoot = root;
        root->peers.push_back(m);
    }
    if (n_devices == 1) { *out = root; return RT_OK; }
    std::vector<rt_context *> members{root};
    members.insert(members.end(), root->peers.begin(), root->peers.end());
    for (rt_context *m : members) {
        hipError_t e = hipSetDevice(m->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->groupDone, hipEventDisableTiming);
        if (e != hipSuccess) { g_create_err = std::string("hipEventCreate: ") + hipGetErrorString(e); rt_destroy(root); return RT_ERR_HIP; }
    }
    if (distinct) {
        std::string why;
        if (!load_rccl(why)) { g_create_err = why; rt_destroy(root); return RT_ERR_UNSUPPORTED; }
        root->comms.assign(n_devices, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(root->comms.data(), n_devices, device_ids);
        if (r != ncclSuccess) {
            g_create_err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
            root->comms.clear();
            rt_destroy(root);
            return RT_ERR_HIP;
        }
    }
    (void)hipSetDevice(root->device);
    *out = root;
    return RT_OK;
}

void rt_shard_range(int spp_begin, int spp_end, int member, int n_members, int *out_begin, int *out_end) {
    // bad arguments (no members, a member outside the group, an inverted range) give the empty range at spp_begin
    long long b = spp_begin, e = spp_begin;
    if (n_members > 0 && member >= 0 && member < n_members && spp_end >= spp_begin) {
        const long long total = (long long)spp_end - spp_begin;
        b = spp_begin + member * total / n_members;
        e = spp_begin + (member + 1) * total / n_members;
    }
    if (out_begin) *out_begin = (int)b;
    if (out_end) *out_end = (int)e;
}

int rt_device_count(const rt_context *c) { return c ? 1 + (int)c->peers.size() : 0; }

void rt_destroy(rt_context *c) {
    if (!c) return;
    for (rt_context *m : c->peers) {
        (void)hipSetDevice(m->device);
        if (m->stream) (void)hipStreamSynchronize(m->stream);
    }
    for (ncclComm_t comm : c->comms)
        if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
    c->comms.clear();
    for (rt_context *m : c->peers) destroy_one(m);
    c->peers.clear();
    destroy_one(c);
}

#define FOR_EACH_MEMBER(c, call)                                                        \
    do {                                                                                \
        if (!(c)) return RT_ERR_INVALID_ARG;                                            \
        rt_context *m = (c);                                                            \
        int rc_ = (call);                                                               \
        if (rc_) return rc_;                                                            \
        for (rt_context *peer_ : (c)->peers) {                                          \
            m = peer_;                                                                  \
            rc_ = (call);                                                               \
            if (rc_) { (c)->err = m->err; (void)hipSetDevice((c)->device); return rc_; }\
        }                                                                               \
        if (!(c)->peers.empty()) (void)hipSetDevice((c)->device);                       \
        return RT_OK;                                                                   \
    } while (0)

int rt_upload_scene(rt_context *c, const rt_scene_desc *d) { FOR_EACH_MEMBER(c, upload_scene_one(m, d)); }
int rt_set_camera(rt_context *c, const float r2c[16], const float c2w[16], int width, int height, const int sb[4]) {
    FOR_EACH_MEMBER(c, set_camera_one(m, r2c, c2w, width, height, sb));
}
int rt_set_film(rt_context *c, float radius_x, float radius_y, const int cb[4], float max_sample_luminance) {
    FOR_EACH_MEMBER(c, set_film_one(m, radius_x, radius_y, cb, max_sample_luminance));
}
int rt_set_option(rt_context *c, int option, int value) { FOR_EACH_MEMBER(c, set_option_one(m, option, value)); }
int rt_reset_stats(rt_context *c) { FOR_EACH_MEMBER(c, reset_stats_one(m)); }
int rt_synchronize(rt_context *c) { FOR_EACH_MEMBER(c, synchronize_one(m)); }

// Counters are summed over the members; times and the state fields are member 0's.
int rt_get_stats(rt_context *c, rt_stats *out) {
    if (!c || !out) return RT_ERR_INVALID_ARG;
    int rc = get_stats_one(c, out);
    for (rt_context *m : c->peers) {
        rt_stats s;
        if (rc == RT_OK && (rc = get_stats_one(m, &s)) == RT_OK) {
            out->rays_closest += s.rays_closest; out->rays_any += s.rays_any; out->camera_samples += s.camera_samples;
            out->n_interior += s.n_interior; out->n_leaf += s.n_leaf; out->n_leaf_idx += s.n_leaf_idx; out->n_tri_tests += s.n_tri_tests;
            out->counted_rays += s.counted_rays; out->extend_launches += s.extend_launches; out->extend_rays += s.extend_rays;
            out->shade_launches += s.shade_launches;
            // kernel times: the slowest member's (the members run side by side); render_ms stays the root's host clock
            out->extend_ms = std::max(out->extend_ms, s.extend_ms);
            out->shade_ms = std::max(out->shade_ms, s.shade_ms);
        } else if (rc) c->err = m->err;
    }
    if (!c->peers.empty()) (void)hipSetDevice(c->device);
    return rc;
}

}  // extern "C"
