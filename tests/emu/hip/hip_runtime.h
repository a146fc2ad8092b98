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
struct EmuStream { int tag; };
struct EmuEvent { int tag; };
typedef EmuStream *hipStream_t;
typedef EmuEvent *hipEvent_t;
struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
};

namespace emu_rt {
inline hipError_t lastError = hipSuccess;
inline int workgroupsPerCU = 2;   // what the occupancy query answers
inline int computeUnits = 2;      // what the device properties say: persistent grids stay small
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
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d) {
    if (d != 0) return hipErrorInvalidDevice;
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "emulated");
    p->multiProcessorCount = emu_rt::computeUnits;
    return hipSuccess;
}
template <typename T> static inline hipError_t hipMalloc(T **p, size_t bytes) {
    *p = static_cast<T *>(std::malloc(bytes ? bytes : 1));   // not cleared: device memory holds whatever it held
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { std::memmove(dst, src, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(dst, src, n); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t = nullptr) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = new EmuStream{1}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new EmuStream{1}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { if (!s || s->tag != 1) return hipErrorInvalidValue; s->tag = 0; delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { return (!s || s->tag == 1) ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent{2}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new EmuEvent{2}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { if (!e || e->tag != 2) return hipErrorInvalidValue; e->tag = 0; delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr) { return (e && e->tag == 2 && (!s || s->tag == 1)) ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { return (e && e->tag == 2) ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { return (e && e->tag == 2 && (!s || s->tag == 1)) ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = 0.f; return (a && b && a->tag == 2 && b->tag == 2) ? hipSuccess : hipErrorInvalidValue; }
template <typename K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = emu_rt::workgroupsPerCU; return hipSuccess; }
template <typename T> static inline hipError_t hipGetSymbolAddress(void **p, T &symbol) { *p = static_cast<void *>(&symbol); return hipSuccess; }

// A launch runs its grid before it returns. A stream argument must be a live stream (or the null stream).
#define hipLaunchKernelGGL(kernel, grid, block, ldsBytes, stream, ...)                                               \
    do {                                                                                                             \
        hipStream_t emu_s_ = (stream);                                                                               \
        if (emu_s_ && emu_s_->tag != 1) simt::fatal("launch on a destroyed stream");                                 \
        const dim3 emu_g_ = (grid), emu_b_ = (block);                                                                \
        if (!simt::run_grid(simt::Dim3{emu_g_.x, emu_g_.y, emu_g_.z}, simt::Dim3{emu_b_.x, emu_b_.y, emu_b_.z}, (size_t)(ldsBytes), \
                            [&]() { kernel(__VA_ARGS__); }))                                                         \
            emu_rt::lastError = hipErrorLaunchFailure;                                                               \
    } while (0)
