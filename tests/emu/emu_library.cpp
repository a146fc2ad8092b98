// The whole device library on the CPU: this translation unit includes raytracing_amd/csrc/device/rt_device.hip unchanged
// (C ABI, host code, kernels) against the stand-ins tests/emu/hip/hip_runtime.h and tests/emu/rccl/rccl.h, and drives it
// through include/raytracing_amd.h the way a host program does. Built with -fsanitize=address, every "device" buffer is a
// heap block of its exact size, every release is a free, every launch runs its grid fibre by fibre before it returns: a
// stale pointer, a buffer one element short, a table index out of range or a barrier that part of a workgroup misses
// ends the run with a report. The test (tests/test_emu_library.py) compares what this program writes with the oracle.
// A checker for the library's source, not a second way to run it: the library itself has no CPU path.
//
//   emu_library <scene.bin> <rays.bin> <out.bin> <devices: 0 | 0,0> <any-hit tMax>
//
// scene.bin / rays.bin as written by tests/flatbin.py (the render job in scene.bin: spp, depth, seed). Sequence:
//   1  rt_create_multi, rt_upload_scene, rt_set_camera
//   2  for the lockstep and the streaming schedule: rt_trace_closest, rt_trace_any            -> out: 2 x {prim, t, bary, occluded}
//   3  rt_render, Path, counter streams                                                        -> out: rgb, w
//   4  rt_render, Whitted on the reference's tile streams, 2 spp (allocates its offset search) -> out: rgb, w
//   5  rt_render as in 3 (buffers of 3 must have survived 4)                                   -> out: rgb, w
//   6  rt_upload_scene again (tables released and rebuilt), rt_trace_closest                   -> out: prim, t, bary
//   7  one device only: rt_render_noalloc_dev, Path as in 3                                    -> out: rgb, w
//   8  rt_develop of 3's film                                                                  -> out: rgb8
//   9  rt_get_stats, rt_destroy
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../raytracing_amd/csrc/device/rt_device.hip"

namespace {

template <typename T>
std::vector<T> read_vec(FILE *f, size_t count) {
    std::vector<T> v(count);
    if (count && fread(v.data(), sizeof(T), count, f) != count) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
    return v;
}

template <typename T>
void write_vec(FILE *f, const std::vector<T> &v) {
    if (!v.empty() && fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) {
        fprintf(stderr, "short write\n");
        exit(2);
    }
}

}   // namespace

#define CHECK(call)                                                                                             \
    do {                                                                                                        \
        const int rc_ = (call);                                                                                 \
        if (rc_ != RT_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rt_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s scene.bin rays.bin out.bin devices anyhit_tmax\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    uint32_t h[7];
    int32_t ih[10];
    uint64_t seed;
    float bounds[6], r2c[16], c2w[16];
    if (fread(h, 4, 7, f) != 7 || h[0] != 0x53465452u || fread(ih, 4, 10, f) != 10 || fread(&seed, 8, 1, f) != 1 || fread(bounds, 4, 6, f) != 6 ||
        fread(r2c, 4, 16, f) != 16 || fread(c2w, 4, 16, f) != 16) {
        fprintf(stderr, "bad scene header\n");
        return 2;
    }
    rt_scene_desc sd = {};
    sd.n_nodes = h[1]; sd.n_leaf_idx = h[2]; sd.n_prims = h[3]; sd.n_materials = h[4]; sd.n_lights = h[5];
    memcpy(sd.bounds, bounds, sizeof bounds);
    // exact-size host arrays: the library must not read past them either
    const std::vector<uint32_t> nodes = read_vec<uint32_t>(f, 2 * (size_t)sd.n_nodes);
    const std::vector<int32_t> leafIdx = read_vec<int32_t>(f, sd.n_leaf_idx);
    const std::vector<float> triP = read_vec<float>(f, 9 * (size_t)sd.n_prims);
    const std::vector<float> triN = h[6] ? read_vec<float>(f, 9 * (size_t)sd.n_prims) : std::vector<float>();
    const std::vector<int32_t> matId = read_vec<int32_t>(f, sd.n_prims), lightId = read_vec<int32_t>(f, sd.n_prims);
    const std::vector<rt_material> mats = read_vec<rt_material>(f, sd.n_materials);
    const std::vector<rt_light> lights = read_vec<rt_light>(f, sd.n_lights);
    fclose(f);
    sd.nodes = nodes.data();
    sd.leaf_idx = leafIdx.empty() ? nullptr : leafIdx.data();
    sd.tri_p = triP.data();
    sd.tri_n = triN.empty() ? nullptr : triN.data();
    sd.material_id = matId.data();
    sd.light_id = lightId.data();
    sd.materials = mats.data();
    sd.lights = lights.empty() ? nullptr : lights.data();

    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) { fprintf(stderr, "bad ray file\n"); return 2; }
    const std::vector<float> o = read_vec<float>(f, 3 * n), d = read_vec<float>(f, 3 * n);
    fclose(f);

    std::vector<int> devices;
    for (const char *p = argv[4]; *p;) {
        devices.push_back((int)strtol(p, (char **)&p, 10));
        if (*p == ',') ++p;
    }
    const std::vector<float> tmaxAny(n, (float)atof(argv[5]));
    FILE *out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 2; }

    rt_context *ctx = nullptr;
    CHECK(rt_create_multi(devices.data(), (int)devices.size(), &ctx));
    CHECK(rt_upload_scene(ctx, &sd));
    const int width = ih[0], height = ih[1];
    CHECK(rt_set_camera(ctx, r2c, c2w, width, height, &ih[2]));

    std::vector<int32_t> prim(n);
    std::vector<float> t(n), bary(3 * n);
    std::vector<uint8_t> occ(n);
    for (int schedule : {RT_SCHEDULE_LOCKSTEP, RT_SCHEDULE_STREAM}) {
        CHECK(rt_set_option(ctx, RT_OPTION_TRACE_SCHEDULE, schedule));
        CHECK(rt_trace_closest(ctx, n, o.data(), d.data(), nullptr, prim.data(), t.data(), bary.data()));
        CHECK(rt_trace_any(ctx, n, o.data(), d.data(), tmaxAny.data(), occ.data()));
        write_vec(out, prim); write_vec(out, t); write_vec(out, bary); write_vec(out, occ);
    }
    CHECK(rt_set_option(ctx, RT_OPTION_TRACE_SCHEDULE, RT_SCHEDULE_AUTO));

    const size_t npix = (size_t)width * height;
    rt_render_params path = {};
    path.integrator = RT_INTEGRATOR_PATH;
    path.spp_total = ih[7];
    path.spp_begin = 0;
    path.spp_end = ih[7];
    path.max_depth = ih[8];
    path.rr_threshold = 1.0f;
    path.rng_mode = RT_RNG_COUNTER;
    path.seed = seed;
    rt_render_params whitted = path;
    whitted.integrator = RT_INTEGRATOR_WHITTED;
    whitted.rng_mode = RT_RNG_REFERENCE_TILE_STREAM;
    whitted.spp_total = whitted.spp_end = 2;
    std::vector<float> film1;
    for (const rt_render_params *job : {&path, &whitted, &path}) {
        std::vector<float> rgb(3 * npix, 0.f), w(npix, 0.f);
        CHECK(rt_render(ctx, job, rgb.data(), w.data()));
        write_vec(out, rgb); write_vec(out, w);
        if (film1.empty()) { film1 = rgb; film1.insert(film1.end(), w.begin(), w.end()); }
    }

    CHECK(rt_upload_scene(ctx, &sd));
    CHECK(rt_trace_closest(ctx, n, o.data(), d.data(), nullptr, prim.data(), t.data(), bary.data()));
    write_vec(out, prim); write_vec(out, t); write_vec(out, bary);

    if (devices.size() == 1) {
        // "device" pointers are heap pointers here; the entry point itself must not allocate (not checked by this program)
        std::vector<float> rgb(3 * npix, 0.f), w(npix, 0.f);
        CHECK(rt_render_noalloc_dev(ctx, &path, rgb.data(), w.data()));
        CHECK(rt_synchronize(ctx));
        write_vec(out, rgb); write_vec(out, w);
    }
    std::vector<uint8_t> rgb8(3 * npix);
    CHECK(rt_develop(ctx, film1.data(), film1.data() + 3 * npix, 1.0f, rgb8.data()));
    write_vec(out, rgb8);
    rt_stats st;
    CHECK(rt_get_stats(ctx, &st));
    printf("%s: %d device(s), %llu rays per batch, %llu camera samples, %llu + %llu render rays, %llu wave-wide operations\n", rt_version(),
           rt_device_count(ctx), (unsigned long long)n, (unsigned long long)st.camera_samples, (unsigned long long)st.rays_closest,
           (unsigned long long)st.rays_any, simt::waveOps);
    rt_destroy(ctx);
    fclose(out);
    return 0;
}
