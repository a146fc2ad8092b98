// Stand-in for <hip/hip_runtime.h> when sources of raytracing_amd/csrc/device are compiled for the CPU by the harnesses
// in tests/emu: the vector types, conversions, atomics and wave intrinsics the kernels use (work-items run as fibres that
// meet at wave-wide operations and workgroup barriers, tests/emu/simt.h), and the part of the HIP runtime API the library's
// host code calls, with device memory on the heap (so AddressSanitizer watches every "device" access and every release),
// streams that run their work at once and in order, and launches that run the grid before they return.
// Test infrastructure: nothing under raytracing_amd/ includes this file.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <set>
#include <tuple>
#include <vector>

#include "../simt.h"

// ---- language -------------------------------------------------------------------------------------------------------
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline   /* forced inlining of the kernels' helpers costs the sanitizer build minutes */
#define __launch_bounds__(...)
#define __shared__ static          /* a workgroup-wide array inside a kernel: one copy for the fibres of the running workgroup */
#define __align__(n) alignas(n)
#define HIP_SYMBOL(x) x
#define RT_DYNAMIC_LDS(name) uint4 *name = reinterpret_cast<uint4 *>(simt::dynamicLds)

// aligned as HIP's vector types are, so that UndefinedBehaviorSanitizer reports a vector access at an address the GPU's wide
// loads and stores could not take
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) uint2 { uint32_t x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

static inline int min(int a, int b) { return b < a ? b : a; }
static inline int max(int a, int b) { return a < b ? b : a; }
static inline unsigned min(unsigned a, unsigned b) { return b < a ? b : a; }
static inline unsigned max(unsigned a, unsigned b) { return a < b ? b : a; }
using std::isinf;
using std::isnan;
#define __lane_id() (simt::current->threadIdx_.x & 63u)

// one thread runs all fibres: an atomic is a plain read-modify-write
template <typename T> static inline T atomicAdd(T *p, T v) { const T old = *p; *p = old + v; return old; }
template <typename T> static inline T atomicOr(T *p, T v) { const T old = *p; *p = old | v; return old; }
template <typename T> static inline T atomicMax(T *p, T v) { const T old = *p; if (v > old) *p = v; return old; }
template <typename T> static inline T atomicMin(T *p, T v) { const T old = *p; if (v < old) *p = v; return old; }

// wave-wide operations and barriers: every live lane that takes part must reach the same one (checked by source line)
#define __ballot(pred) simt::ballot((pred), __LINE__)
#define RT_BALLOT(pred) simt::ballot((pred), __LINE__)
#define RT_MUL24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#define __builtin_amdgcn_wave_barrier() simt::wave_barrier(__LINE__)
#define __syncthreads() simt::block_barrier(__LINE__)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_readfirstlane(v) (v)   /* only used on wave-uniform values */

#define threadIdx (simt::current->threadIdx_)
#define blockIdx (simt::launch.blockIdx_)
#define blockDim (simt::launch.blockDim_)
#define gridDim (simt::launch.gridDim_)

// the register pin of dtraverse.h is an AMD GPU constraint; on the CPU there is nothing to keep apart
#define RT_PIN_VGPRS(a, b) ((void)0)

// ---- runtime --------------------------------------------------------------------------------------------------------
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorLaunchFailure = 719 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
};

// Streams and events. Two schedules, chosen when the process starts:
//   eager (default): work runs when it is queued, so every stream order is program order;
//   lazy (EMU_LAZY_STREAMS=1): work stays queued until somebody needs it - a stream synchronize, an event another stream
//     waits for, a release of device memory - and then only what is needed runs, in dependency order. Work on different
//     streams is thereby reordered as far as the events allow: a copy that reads a buffer before its producer's stream was
//     waited for, or a kernel that needs a fill queued on another stream without an event between them, sees stale data
//     and the comparison with the oracle fails. The null stream is a stream of its own that the library's non-blocking
//     streams do not wait for, as on the device.
struct EmuEvent {
    int tag;
    unsigned long long recorded = 0;   // number of the last hipEventRecord call on this event (calls are numbered process-wide)
};
struct EmuItem {
    enum Kind { Work, Record, Wait } kind;
    std::function<void()> fn;
    EmuEvent *event;
    unsigned long long seq;
};
struct EmuStream {
    int tag;
    std::deque<EmuItem> pending;
    bool draining = false;
};
typedef EmuStream *hipStream_t;
typedef EmuEvent *hipEvent_t;

namespace emu_rt {
inline hipError_t lastError = hipSuccess;
inline int workgroupsPerCU = 2;   // what the occupancy query answers
inline int computeUnits = 2;      // what the device properties say: persistent grids stay small
inline int devices = 2;           // device memory is the heap for all of them; a stream does not remember its device
inline const bool lazy = std::getenv("EMU_LAZY_STREAMS") != nullptr;
inline EmuStream nullStream{1};
inline std::vector<EmuStream *> streams{&nullStream};
inline unsigned long long recordCalls = 0;
inline std::vector<EmuEvent *> retired;   // events the RCCL stand-in created; released once every stream is empty
inline std::set<unsigned long long> recordsDone;   // record calls that have run: a wait names one call, not the event's latest
inline bool done(unsigned long long seq) { return seq == 0 || recordsDone.count(seq) != 0; }

inline EmuStream *real(hipStream_t s) { return s ? s : &nullStream; }
inline void run_until(EmuStream *s, unsigned long long seq);

// Runs the front item of `s`; an event wait first runs the recording stream up to that record.
inline void run_front(EmuStream *s) {
    EmuItem it = std::move(s->pending.front());
    s->pending.pop_front();
    if (it.kind == EmuItem::Work) {
        it.fn();
    } else if (it.kind == EmuItem::Record) {
        recordsDone.insert(it.seq);
    } else if (!done(it.seq)) {
        EmuStream *producer = nullptr;
        for (EmuStream *p : streams)
            for (const EmuItem &q : p->pending)
                if (q.kind == EmuItem::Record && q.seq == it.seq) producer = p;
        if (!producer) simt::fatal("a stream waits for an event record that is not queued anywhere");
        if (producer->draining) simt::fatal("streams wait for each other in a cycle");
        run_until(producer, it.seq);
    }
}
// Runs `s` until record call `seq` has run (0: until the stream is empty).
inline void run_until(EmuStream *s, unsigned long long seq) {
    s->draining = true;
    while (!s->pending.empty() && !(seq && done(seq))) run_front(s);
    s->draining = false;
}
inline void drain(EmuStream *s) { run_until(s, 0); }
inline void drain_all() {
    for (size_t i = 0; i < streams.size(); ++i) drain(streams[i]);
    for (EmuEvent *e : retired) delete e;
    retired.clear();
}
inline void submit(hipStream_t stream, EmuItem item) {
    EmuStream *s = real(stream);
    if (s->tag != 1) simt::fatal("work queued on a destroyed stream");
    s->pending.push_back(std::move(item));
    if (!lazy) drain(s);
}
}   // namespace emu_rt

static inline const char *hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorInvalidValue: return "invalid value";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorInvalidDevice: return "invalid device";
        case hipErrorLaunchFailure: return "the emulated launch failed (see stderr)";
    }
    return "unknown error";
}
static inline hipError_t hipGetLastError() { const hipError_t e = emu_rt::lastError; emu_rt::lastError = hipSuccess; return e; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = emu_rt::devices; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d >= 0 && d < emu_rt::devices ? hipSuccess : hipErrorInvalidDevice; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d) {
    if (d < 0 || d >= emu_rt::devices) return hipErrorInvalidDevice;
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "emulated");
    p->multiProcessorCount = emu_rt::computeUnits;
    return hipSuccess;
}
template <typename T> static inline hipError_t hipMalloc(T **p, size_t bytes) {
    *p = static_cast<T *>(std::malloc(bytes ? bytes : 1));   // not cleared: device memory holds whatever it held
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
// hipFree waits for the device; the host-blocking copies and hipMemset run on the null stream.
static inline hipError_t hipFree(void *p) { emu_rt::drain_all(); std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) {
    emu_rt::drain(&emu_rt::nullStream);
    std::memmove(dst, src, n);
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s = nullptr) {
    emu_rt::submit(s, EmuItem{EmuItem::Work, [=]() { std::memmove(dst, src, n); }, nullptr, 0});
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s = nullptr) {
    emu_rt::submit(s, EmuItem{EmuItem::Work, [=]() { std::memset(p, v, n); }, nullptr, 0});
    return hipSuccess;
}
static inline hipError_t hipMemset(void *p, int v, size_t n) { return hipMemsetAsync(p, v, n, nullptr); }   // asynchronous for device memory
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags) {
    if (!(flags & hipStreamNonBlocking)) simt::fatal("only non-blocking streams are emulated");
    *s = new EmuStream{1};
    emu_rt::streams.push_back(*s);
    return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t *) { simt::fatal("only non-blocking streams are emulated"); }
static inline hipError_t hipStreamSynchronize(hipStream_t s) {
    if (s && s->tag != 1) return hipErrorInvalidValue;
    emu_rt::drain(emu_rt::real(s));
    return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t s) {
    if (!s || s->tag != 1) return hipErrorInvalidValue;
    emu_rt::drain(s);
    for (size_t i = 0; i < emu_rt::streams.size(); ++i)
        if (emu_rt::streams[i] == s) emu_rt::streams.erase(emu_rt::streams.begin() + (long)i);
    s->tag = 0;
    delete s;
    return hipSuccess;
}
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent{2}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new EmuEvent{2}; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) {
    if (!e || e->tag != 2) return hipErrorInvalidValue;
    if (!emu_rt::done(e->recorded)) {
        EmuStream *producer = nullptr;
        for (EmuStream *p : emu_rt::streams)
            for (const EmuItem &q : p->pending)
                if (q.kind == EmuItem::Record && q.seq == e->recorded) producer = p;
        if (!producer) simt::fatal("an event's record is not queued anywhere");
        emu_rt::run_until(producer, e->recorded);
    }
    return hipSuccess;
}
static inline hipError_t hipEventDestroy(hipEvent_t e) {
    if (!e || e->tag != 2) return hipErrorInvalidValue;
    hipEventSynchronize(e);
    for (EmuStream *p : emu_rt::streams)
        for (const EmuItem &q : p->pending)
            if (q.event == e) simt::fatal("an event is destroyed while a stream still waits for it");
    e->tag = 0;
    delete e;
    return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr) {
    if (!e || e->tag != 2 || (s && s->tag != 1)) return hipErrorInvalidValue;
    e->recorded = ++emu_rt::recordCalls;
    emu_rt::submit(s, EmuItem{EmuItem::Record, nullptr, e, e->recorded});
    return hipSuccess;
}
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    if (!e || e->tag != 2 || (s && s->tag != 1)) return hipErrorInvalidValue;
    if (e->recorded) emu_rt::submit(s, EmuItem{EmuItem::Wait, nullptr, e, e->recorded});   // an event never recorded holds nobody up
    return hipSuccess;
}
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    if (!a || !b || a->tag != 2 || b->tag != 2) return hipErrorInvalidValue;
    if (!emu_rt::done(a->recorded) || !emu_rt::done(b->recorded)) return hipErrorInvalidValue;   // hipErrorNotReady on the device
    *ms = 0.f;
    return hipSuccess;
}
template <typename K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = emu_rt::workgroupsPerCU; return hipSuccess; }
template <typename T> static inline hipError_t hipGetSymbolAddress(void **p, T &symbol) { *p = static_cast<void *>(&symbol); return hipSuccess; }

// A launch copies its arguments when it is queued, as the device runtime does, and runs its grid when its stream gets there.
namespace emu_rt {
template <typename K, typename... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t ldsBytes, hipStream_t stream, Args... args) {
    auto bound = std::make_tuple(args...);
    submit(stream, EmuItem{EmuItem::Work, [=]() {
                               if (!simt::run_grid(simt::Dim3{grid.x, grid.y, grid.z}, simt::Dim3{block.x, block.y, block.z}, ldsBytes,
                                                   [&]() { std::apply(kernel, bound); }))
                                   lastError = hipErrorLaunchFailure;
                           },
                           nullptr, 0});
}
}   // namespace emu_rt
#define hipLaunchKernelGGL(kernel, grid, block, ldsBytes, stream, ...) \
    emu_rt::launch((kernel), dim3(grid), dim3(block), (size_t)(ldsBytes), (stream), ##__VA_ARGS__)
