// Stand-in for <hip/hip_runtime.h> when the device headers of raytracing_amd/csrc/device are compiled for the CPU by
// tests/emu/emu_trace.cpp: the few vector types, conversions and wave intrinsics those headers use, with one wave of 64
// lanes run as 64 fibres that meet at every wave-wide operation (tests/emu/simt.h). Test infrastructure: nothing under
// raytracing_amd/ includes this file.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../simt.h"

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

// wave-wide operations: every live lane of the wave must reach the same one (checked by source line)
#define __ballot(pred) simt::ballot((pred), __LINE__)
#define __builtin_amdgcn_wave_barrier() simt::wave_barrier(__LINE__)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_readfirstlane(v) (v)   /* only used on wave-uniform values */
#define __syncthreads() simt::unsupported("__syncthreads")

#define threadIdx (simt::current->threadIdx_)
#define blockIdx (simt::launch.blockIdx_)
#define blockDim (simt::launch.blockDim_)
#define gridDim (simt::launch.gridDim_)

// the register pin of dtraverse.h is an AMD GPU constraint; on the CPU there is nothing to keep apart
#define RT_PIN_VGPRS(a, b) ((void)0)
