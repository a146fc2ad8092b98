// Device-side float math. Every expression keeps the association order of the scalar glm paths the
// reference evaluates on the CPU (reference src/vendor/header_only/glm/detail/func_geometric.inl:48-90),
// and this file is compiled with -ffp-contract=off, so +,-,*,/ and sqrt round exactly as they do there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The workgroup's dynamic LDS (the size given at launch) as an array of uint4.
#ifndef RT_DYNAMIC_LDS
#define RT_DYNAMIC_LDS(name) extern __shared__ uint4 name[]
#endif

// Wave-wide vote on a bool (the compare's own lane mask: no conversion of the predicate to an integer and back), and a
// 24-bit multiply (full rate; a 32-bit one takes four issue slots).
#ifndef RT_BALLOT
#define RT_BALLOT(pred) __builtin_amdgcn_ballot_w64(pred)
#endif
#ifndef RT_MUL24
#define RT_MUL24(a, b) __umul24((a), (b))
#endif

#define RT_DEV __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__

struct f3 {
    float x, y, z;
};
RT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD float absdot3(f3 a, f3 b) { return fabsf(dot3(a, b)); }
RT_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
RT_HD float length3(f3 a) { return sqrtf(dot3(a, a)); }   // glm::length
RT_HD f3 normalize3(f3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
RT_HD float gmin(float x, float y) { return (y < x) ? y : x; }   // glm::min: NaN handling differs from fminf
RT_HD float gmax(float x, float y) { return (x < y) ? y : x; }   // glm::max
RT_HD float comp3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
RT_HD bool black3(f3 c) { return c.x == 0.f && c.y == 0.f && c.z == 0.f; }

#define RT_PI 3.14159265358979323846f
#define RT_INV_PI 0.31830988618379067154f
#define RT_PI_OVER_2 1.57079632679489661923f
#define RT_PI_OVER_4 0.78539816339744830961f
#define RT_ONE_MINUS_EPS 0.99999994f
#define RT_SHADOW_EPS 0.0001f
#define RT_INF __builtin_huge_valf()
#define RT_MACH_EPS 5.9604644775390625e-08f   // 2^-24 = numeric_limits<float>::epsilon() * 0.5

// gamma(n) = (n * eps) / (1 - n * eps) (reference src/core/Utils/Base.h:48), evaluated in float like there.
RT_DEV float gamma_err(int n) { return ((float)n * RT_MACH_EPS) / (1.0f - (float)n * RT_MACH_EPS); }

// PCG32 (reference src/core/Utils/Math.h:690-790).
struct Pcg {
    uint64_t state, inc;
};
RT_DEV uint32_t pcg_next(Pcg &r) {
    uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
RT_DEV void pcg_set_sequence(Pcg &r, uint64_t seq) {
    r.state = 0u;
    r.inc = (seq << 1u) | 1u;
    pcg_next(r);
    r.state += 0x853c49e6748fea9bULL;
    pcg_next(r);
}
// Rng::advance (reference src/core/Utils/Math.h:730-746): jumps the stream forward by `delta` draws in O(log delta).
RT_DEV void pcg_advance(Pcg &r, uint64_t delta) {
    uint64_t curMult = 0x5851f42d4c957f2dULL, curPlus = r.inc, accMult = 1u, accPlus = 0u;
    while (delta > 0) {
        if (delta & 1) {
            accMult *= curMult;
            accPlus = accPlus * curMult + curPlus;
        }
        curPlus = (curMult + 1) * curPlus;
        curMult *= curMult;
        delta /= 2;
    }
    r.state = accMult * r.state + accPlus;
}
RT_DEV float pcg_float(Pcg &r) { return gmin(RT_ONE_MINUS_EPS, (float)pcg_next(r) * 2.3283064365386963e-10f); }
RT_DEV uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// Stream of camera sample `s` of pixel `pix` in counter mode.
RT_DEV uint64_t counter_sequence(uint64_t pix, uint32_t spp_total, uint32_t s, uint64_t seed) {
    return splitmix64(pix * (uint64_t)spp_total + (uint64_t)s + seed * 0x9E3779B97F4A7C15ULL);
}

// sin and cos of an angle in [-pi/4 - eps, 3pi/4 + eps] (the only range the concentric disk mapping
// produces, reference src/core/Render/Sampler.cpp:120-142), evaluated in double and rounded once to
// float: agrees with a correctly rounded sinf/cosf except when the exact value lies within ~1e-9 ulp
// of a rounding boundary.
RT_HD void sincos_disk(float theta, float &s, float &c) {
    double t = (double)theta;
    bool upper = theta > RT_PI_OVER_4;
    double r = upper ? t - 1.5707963267948966192313216916398 : t;   // |r| <= pi/4 (+ float slack)
    double r2 = r * r;
    // Taylor polynomials to r^17 / r^16: truncation error < 2e-17 on |r| <= 0.79
    double sp = -2.8114572543455206e-15;                 // -1/17!
    sp = __builtin_fma(sp, r2, 7.6471637318198164e-13);  // 1/15!
    sp = __builtin_fma(sp, r2, -1.6059043836821613e-10); // -1/13!
    sp = __builtin_fma(sp, r2, 2.5052108385441720e-08);  // 1/11!
    sp = __builtin_fma(sp, r2, -2.7557319223985893e-06); // -1/9!
    sp = __builtin_fma(sp, r2, 1.9841269841269841e-04);  // 1/7!
    sp = __builtin_fma(sp, r2, -8.3333333333333332e-03); // -1/5!
    sp = __builtin_fma(sp, r2, 1.6666666666666666e-01);  // 1/3!  (sign applied below)
    double sr = r - r * r2 * sp;
    double cp = 4.7794773323873853e-14;                  // 1/16!
    cp = __builtin_fma(cp, r2, -1.1470745597729725e-11); // -1/14!
    cp = __builtin_fma(cp, r2, 2.0876756987868100e-09);  // 1/12!
    cp = __builtin_fma(cp, r2, -2.7557319223985888e-07); // -1/10!
    cp = __builtin_fma(cp, r2, 2.4801587301587302e-05);  // 1/8!
    cp = __builtin_fma(cp, r2, -1.3888888888888889e-03); // -1/6!
    cp = __builtin_fma(cp, r2, 4.1666666666666664e-02);  // 1/4!
    cp = __builtin_fma(cp, r2, -0.5);
    double cr = __builtin_fma(cp, r2, 1.0);
    // theta = r + pi/2: sin(theta) = cos(r), cos(theta) = -sin(r)
    s = (float)(upper ? cr : sr);
    c = (float)(upper ? -sr : cr);
}

// sin and cos of any angle in [-2 pi, 2 pi], same evaluation (quadrant reduction with a two-part pi/2, then the
// polynomials above): used where the reference calls sinf / cosf on an azimuth or a polar angle.
RT_HD void sincos_cr(float theta, float &s, float &c) {
    const double t = (double)theta;
    const double kd = __builtin_rint(t * 0.63661977236758134308);   // t / (pi/2)
    const int k = (int)kd;
    double r = (t - kd * 1.57079632679489655800e+00) - kd * 6.12323399573676603587e-17;
    double r2 = r * r;
    double sp = -2.8114572543455206e-15;
    sp = __builtin_fma(sp, r2, 7.6471637318198164e-13);
    sp = __builtin_fma(sp, r2, -1.6059043836821613e-10);
    sp = __builtin_fma(sp, r2, 2.5052108385441720e-08);
    sp = __builtin_fma(sp, r2, -2.7557319223985893e-06);
    sp = __builtin_fma(sp, r2, 1.9841269841269841e-04);
    sp = __builtin_fma(sp, r2, -8.3333333333333332e-03);
    sp = __builtin_fma(sp, r2, 1.6666666666666666e-01);
    const double sr = r - r * r2 * sp;
    double cp = 4.7794773323873853e-14;
    cp = __builtin_fma(cp, r2, -1.1470745597729725e-11);
    cp = __builtin_fma(cp, r2, 2.0876756987868100e-09);
    cp = __builtin_fma(cp, r2, -2.7557319223985888e-07);
    cp = __builtin_fma(cp, r2, 2.4801587301587302e-05);
    cp = __builtin_fma(cp, r2, -1.3888888888888889e-03);
    cp = __builtin_fma(cp, r2, 4.1666666666666664e-02);
    cp = __builtin_fma(cp, r2, -0.5);
    const double cr = __builtin_fma(cp, r2, 1.0);
    const int q = k & 3;
    const double sv = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    const double cv = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
    s = (float)sv;
    c = (float)cv;
}

// acos rounded once from double (the reference calls acosf on the cosine of a sphere hit's polar angle).
RT_HD float acos_cr(float x) { return (float)acos((double)x); }

