// The tracing kernels' device code on the CPU: compiles raytracing_amd/csrc/device/dstream.h and dtraverse.h unchanged
// against the stand-in tests/emu/hip/hip_runtime.h, runs the streaming tracer or the lockstep tracer, in the flavour for
// tables in HBM or for tables in LDS, over a ray batch one emulated wave at a time, and writes ids / t / barycentrics /
// occlusion flags for the test to compare with the oracle. What this checks that a GPU run cannot show as clearly: every
// table and stack index stays in range (build with -fsanitize=address: the tables are heap blocks of their exact size plus
// the 16 bytes of padding the library's upload adds), every wave-wide operation is reached by all live lanes together, and
// every wave ends.
//
//   emu_trace <scene.bin> <rays.bin> <out.bin> <lockstep|stream|refill>[-lds] <blocks> <any-hit tMax>
//
// The -lds suffix selects the flavour for scenes whose tables fit in LDS: three copies of the triangle table, one per axis
// permutation of the triangle test, and the short LDS part of the todo stack.
// scene.bin / rays.bin as written by tests/flatbin.py. out.bin: int32 prim[n], float t[n], float bary[3n], uint8 occluded[n].
// Build flag -DRT_STREAM_MERGED=1 selects the merged-round flavour of the streaming tracer (tables in HBM only).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../raytracing_amd/csrc/device/dstream.h"
#include "../../raytracing_amd/csrc/device/drefill.h"

namespace {

template <typename T>
std::vector<T> read_array(FILE *f, size_t count) {
    std::vector<T> v(count);
    if (count && fread(v.data(), sizeof(T), count, f) != count) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
    return v;
}

// A device table: exact size plus the 16 zero bytes that the library's upload appends.
template <typename T>
T *device_table(const std::vector<T> &v) {
    const size_t bytes = v.size() * sizeof(T);
    const size_t alloc = (bytes + 15) / 16 * 16 + 16;
    char *p = (char *)calloc(alloc, 1);
    if (bytes) memcpy(p, v.data(), bytes);
    return (T *)p;
}

struct Batch {
    uint64_t n;
    const float *o, *d;
    float anyTMax;
    int *prim;
    float *t, *bary;
    const float *tmaxAny;
    uint8_t *occ;
};

// One launch of `blocks` workgroups, wave after wave. LDS_SCENE picks the template flavour exactly as the kernels do.
template <bool LDS_SCENE>
bool run_grid(const DevScene &sc, const SceneTables &tb, const Batch &b, bool stream, bool refill, bool closest, unsigned blocks, unsigned long long maxOps) {
    constexpr int SL = StackCfg<LDS_SCENE>::LDS_N;
    // the workgroup's LDS: todo stack part and prepared-ray pool, exact sizes
    std::vector<uint32_t> ldsStack((size_t)SL * 2 * RT_BLOCK);
    std::vector<uint4> ldsPool((size_t)RT_BLOCK * RT_POOL_VEC_PER_RAY);
    for (unsigned blk = 0; blk < blocks; ++blk) {
        simt::launch.blockIdx_ = simt::Dim3{blk, 0, 0};
        for (unsigned w = 0; w < RT_BLOCK / 64; ++w) {
            bool ok;
            if (refill) {
                ok = simt::run_wave(w, [&]() {
                    TravCounters cn = {0, 0, 0, 0, 0};
                    if (closest) {
                        TraceClosestJob job = {b.o, b.d, nullptr, b.prim, b.t, b.bary};
                        trace_refill<true, false, SL, LDS_SCENE, false>(sc, tb, ldsStack.data(), (uint32_t)b.n, job, cn);
                    } else {
                        TraceAnyJob job = {b.o, b.d, b.tmaxAny, b.occ};
                        trace_refill<false, false, SL, LDS_SCENE, false>(sc, tb, ldsStack.data(), (uint32_t)b.n, job, cn);
                    }
                }, maxOps);
            } else if (stream) {
                ok = simt::run_wave(w, [&]() {
                    TravCounters cn = {0, 0, 0, 0, 0};
                    if (closest) {
                        TraceClosestJob job = {b.o, b.d, nullptr, b.prim, b.t, b.bary};
                        trace_stream<true, false, SL, LDS_SCENE, false>(sc, tb, ldsStack.data(), ldsPool.data(), (uint32_t)b.n, job, cn);
                    } else {
                        TraceAnyJob job = {b.o, b.d, b.tmaxAny, b.occ};
                        trace_stream<false, false, SL, LDS_SCENE, false>(sc, tb, ldsStack.data(), ldsPool.data(), (uint32_t)b.n, job, cn);
                    }
                }, maxOps);
            } else {
                ok = simt::run_wave(w, [&]() {
                    TravCounters cn = {0, 0, 0, 0, 0};
                    for (uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < b.n; i += (uint64_t)gridDim.x * RT_BLOCK) {
                        RayCtx r;
                        ray_setup(r, mk3(b.o[3 * i], b.o[3 * i + 1], b.o[3 * i + 2]), mk3(b.d[3 * i], b.d[3 * i + 1], b.d[3 * i + 2]));
                        HitRec hr;
                        if (closest) {
                            kd_trace<true, false, SL, LDS_SCENE, false>(sc, tb, ldsStack.data(), threadIdx.x, r, RT_INF, hr, cn);
                            b.prim[i] = hr.prim;
                            b.t[i] = hr.t;
                            b.bary[3 * i] = hr.b0; b.bary[3 * i + 1] = hr.b1; b.bary[3 * i + 2] = hr.b2;
                        } else {
                            b.occ[i] = kd_trace<false, false, SL, LDS_SCENE, false>(sc, tb, ldsStack.data(), threadIdx.x, r, b.anyTMax, hr, cn) ? 1 : 0;
                        }
                    }
                }, maxOps);
            }
            if (!ok) return false;
        }
    }
    return true;
}

}   // namespace

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s scene.bin rays.bin out.bin lockstep|stream[-lds] blocks anyhit_tmax\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    uint32_t h[7];
    int32_t ih[10];
    uint64_t seed;
    float bounds[6], camera[32];   // raster_to_camera and camera_to_world: not used here
    if (fread(h, 4, 7, f) != 7 || h[0] != 0x53465452u || fread(ih, 4, 10, f) != 10 || fread(&seed, 8, 1, f) != 1 || fread(bounds, 4, 6, f) != 6 ||
        fread(camera, 4, 32, f) != 32) {
        fprintf(stderr, "bad scene header\n");
        return 2;
    }
    const uint32_t nNodes = h[1], nLeafIdx = h[2], nPrims = h[3];
    const std::vector<uint2> nodes = read_array<uint2>(f, nNodes);
    const std::vector<int> leafIdx = read_array<int>(f, nLeafIdx);
    const std::vector<float> triP = read_array<float>(f, 9 * (size_t)nPrims);
    fclose(f);
    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) { fprintf(stderr, "bad ray file\n"); return 2; }
    const std::vector<float> o = read_array<float>(f, 3 * n), d = read_array<float>(f, 3 * n);
    fclose(f);
    const std::string mode = argv[4];
    const bool stream = mode.rfind("stream", 0) == 0;
    const bool refill = mode.rfind("refill", 0) == 0;
    const bool ldsScene = mode.size() > 4 && mode.compare(mode.size() - 4, 4, "-lds") == 0;
    const unsigned blocks = (unsigned)atoi(argv[5]);
    const float anyTMax = (float)atof(argv[6]);

    // triHot as rt_upload_scene lays it out: w of p0 = 1 for a triangle without area; a scene in LDS carries three copies,
    // block b > 0 with the coordinates permuted for kz = b - 1 (x' = coordinate b mod 3, y' the next one, z' = coordinate kz)
    const uint32_t triCopies = ldsScene ? 3u : 1u;
    std::vector<float4> hot(3 * (size_t)nPrims * triCopies);
    for (uint32_t i = 0; i < nPrims; ++i) {
        const float *p = &triP[9 * (size_t)i];
        const f3 p0 = mk3(p[0], p[1], p[2]), p1 = mk3(p[3], p[4], p[5]), p2 = mk3(p[6], p[7], p[8]);
        const f3 g = cross3(p2 - p0, p1 - p0);
        const float flag = dot3(g, g) == 0 ? 1.f : 0.f;
        for (uint32_t b = 0; b < triCopies; ++b) {
            const int kx = b == 0 ? 0 : (int)b % 3, ky = b == 0 ? 1 : (kx + 1) % 3, kz = b == 0 ? 2 : (int)b - 1;
            for (int v = 0; v < 3; ++v)
                hot[((size_t)b * nPrims + i) * 3 + v] = make_float4(p[3 * v + kx], p[3 * v + ky], p[3 * v + kz], v == 0 ? flag : 0.f);
        }
    }
    DevScene sc = {};
    sc.nodes = device_table(nodes);
    sc.leafIdx = device_table(leafIdx);
    sc.triHot = device_table(hot);
    for (int k = 0; k < 3; ++k) { sc.bmin[k] = bounds[k]; sc.bmax[k] = bounds[3 + k]; }
    sc.nNodes = nNodes;
    sc.nLeafIdx = nLeafIdx;
    sc.nPrims = nPrims;
    sc.triCopies = triCopies;
    const SceneTables tb = {sc.nodes, sc.leafIdx, sc.triHot};

    std::vector<int> prim(n, -2);
    std::vector<float> t(n, -1.f), bary(3 * n, -1.f), tmaxAny(n, anyTMax);
    std::vector<uint8_t> occ(n, 255);
    simt::launch.blockDim_ = simt::Dim3{RT_BLOCK, 1, 1};
    simt::launch.gridDim_ = simt::Dim3{blocks, 1, 1};
    const unsigned long long maxOps = 64ull * 1000 * 1000;
    const Batch batch = {n, o.data(), d.data(), anyTMax, prim.data(), t.data(), bary.data(), tmaxAny.data(), occ.data()};
    for (int closest = 1; closest >= 0; --closest) {
        const bool ok = ldsScene ? run_grid<true>(sc, tb, batch, stream, refill, closest != 0, blocks, maxOps)
                                 : run_grid<false>(sc, tb, batch, stream, refill, closest != 0, blocks, maxOps);
        if (!ok) return 3;
    }
    for (uint64_t i = 0; i < n; ++i)
        if (prim[i] == -2 || occ[i] == 255) {
            fprintf(stderr, "ray %llu was never finished\n", (unsigned long long)i);
            return 4;
        }
    f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 2; }
    fwrite(prim.data(), 4, n, f);
    fwrite(t.data(), 4, n, f);
    fwrite(bary.data(), 4, 3 * n, f);
    fwrite(occ.data(), 1, n, f);
    fclose(f);
    printf("%llu rays, %u blocks, %s%s, %llu wave-wide operations\n", (unsigned long long)n, blocks, argv[4],
           (stream && !ldsScene && RT_STREAM_MERGED) ? " (merged rounds)" : "", simt::waveOps);
    if (getenv("EMU_OPS_BY_LINE"))
        for (const auto &kv : simt::opsAtLine)
            printf("  dstream.h:%d%s  %llu  mean lanes with a true predicate %.1f\n", kv.first < 0 ? -kv.first : kv.first, kv.first < 0 ? " (barrier)" : "",
                   kv.second, (double)simt::lanesAtLine[kv.first] / (double)kv.second);
    free((void *)sc.nodes); free((void *)sc.leafIdx); free((void *)sc.triHot);
    return 0;
}
