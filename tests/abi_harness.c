/* A caller of the C ABI with no Python in the process: what a C/C++ host (the reference's Renderer subclass,
 * src/core/Render/Render.h:20-21) does with include/raytracing_amd.h. Reads a flattened scene and a ray batch from flat
 * binary files (written by tests/flatbin.py from a fixture), uploads the scene, traces the batch with
 * rt_trace_closest / rt_trace_any, renders with rt_render, develops the film with rt_develop, and writes every output to
 * one flat binary file that the test compares with the ctypes path.
 *
 *   abi_harness <scene.bin> <rays.bin> <out.bin> [device list, e.g. 0 or 0,0]
 *
 * scene.bin: uint32 magic 'RTFS', then uint32 n_nodes, n_leaf_idx, n_prims, n_materials, n_lights, has_normals,
 *            int32 width, height, sample_bounds[4], integrator, spp, max_depth, rng_mode, uint64 seed,
 *            float bounds[6], raster_to_camera[16], camera_to_world[16], then the arrays of rt_scene_desc in the order
 *            nodes, leaf_idx, tri_p, tri_n (if has_normals), material_id, light_id, materials, lights.
 * rays.bin:  uint64 n, float o[3n], float d[3n].
 * out.bin:   int32 prim[n], float t[n], float bary[3n], uint8 occluded[n], float rgb_sum[3 W H], float weight_sum[W H],
 *            uint8 rgb8[3 W H]. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raytracing_amd.h"

static void *read_array(FILE *f, size_t count, size_t size) {
    void *p = malloc(count * size + 16);
    if (!p || (count && fread(p, size, count, f) != count)) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

#define CHECK(call)                                                                           \
    do {                                                                                      \
        int rc_ = (call);                                                                     \
        if (rc_ != RT_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rt_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s scene.bin rays.bin out.bin [devices]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    uint32_t h[7];
    int32_t ih[10];
    uint64_t seed;
    float bounds[6], r2c[16], c2w[16];
    if (fread(h, 4, 7, f) != 7 || h[0] != 0x53465452u || fread(ih, 4, 10, f) != 10 || fread(&seed, 8, 1, f) != 1 ||
        fread(bounds, 4, 6, f) != 6 || fread(r2c, 4, 16, f) != 16 || fread(c2w, 4, 16, f) != 16) {
        fprintf(stderr, "bad scene header\n");
        return 2;
    }
    rt_scene_desc sd;
    memset(&sd, 0, sizeof sd);
    sd.n_nodes = h[1]; sd.n_leaf_idx = h[2]; sd.n_prims = h[3]; sd.n_materials = h[4]; sd.n_lights = h[5];
    memcpy(sd.bounds, bounds, sizeof bounds);
    sd.nodes = (const uint32_t *)read_array(f, 2 * (size_t)sd.n_nodes, 4);
    sd.leaf_idx = (const int32_t *)read_array(f, sd.n_leaf_idx, 4);
    sd.tri_p = (const float *)read_array(f, 9 * (size_t)sd.n_prims, 4);
    if (h[6]) sd.tri_n = (const float *)read_array(f, 9 * (size_t)sd.n_prims, 4);
    sd.material_id = (const int32_t *)read_array(f, sd.n_prims, 4);
    sd.light_id = (const int32_t *)read_array(f, sd.n_prims, 4);
    sd.materials = (const rt_material *)read_array(f, sd.n_materials, sizeof(rt_material));
    sd.lights = (const rt_light *)read_array(f, sd.n_lights, sizeof(rt_light));
    fclose(f);

    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) { fprintf(stderr, "bad ray file\n"); return 2; }
    float *o = (float *)read_array(f, 3 * n, 4), *d = (float *)read_array(f, 3 * n, 4);
    fclose(f);

    int devices[64], n_devices = 0;
    const char *list = argc > 4 ? argv[4] : "0";
    for (const char *p = list; *p && n_devices < 64;) {
        devices[n_devices++] = (int)strtol(p, (char **)&p, 10);
        if (*p == ',') ++p;
    }

    rt_context *ctx = NULL;
    CHECK(rt_create_multi(devices, n_devices, &ctx));
    CHECK(rt_upload_scene(ctx, &sd));
    /* the library copied the scene: the host's arrays may go, as the reference's loader frees its own */
    free((void *)sd.nodes); free((void *)sd.leaf_idx); free((void *)sd.tri_p); free((void *)sd.tri_n);
    free((void *)sd.material_id); free((void *)sd.light_id); free((void *)sd.materials); free((void *)sd.lights);
    const int width = ih[0], height = ih[1];
    CHECK(rt_set_camera(ctx, r2c, c2w, width, height, &ih[2]));

    int32_t *prim = (int32_t *)malloc(n * 4 + 16);
    float *t = (float *)malloc(n * 4 + 16), *bary = (float *)malloc(n * 12 + 16);
    uint8_t *occ = (uint8_t *)malloc(n + 16);
    CHECK(rt_trace_closest(ctx, n, o, d, NULL, prim, t, bary));
    CHECK(rt_trace_any(ctx, n, o, d, NULL, occ));

    const size_t npix = (size_t)width * height;
    float *rgb = (float *)calloc(npix * 3, 4), *w = (float *)calloc(npix, 4);
    uint8_t *rgb8 = (uint8_t *)malloc(npix * 3);
    rt_render_params rp;
    memset(&rp, 0, sizeof rp);
    rp.integrator = ih[6];
    rp.spp_total = ih[7];
    rp.spp_begin = 0;
    rp.spp_end = ih[7];
    rp.max_depth = ih[8];
    rp.rr_threshold = 1.0f;
    rp.rng_mode = ih[9];
    rp.seed = seed;
    CHECK(rt_render(ctx, &rp, rgb, w));
    CHECK(rt_develop(ctx, rgb, w, 1.0f, rgb8));
    rt_stats st;
    CHECK(rt_get_stats(ctx, &st));
    printf("%s: %d device(s), %llu rays traced twice, %llu camera samples, %llu + %llu render rays\n", rt_version(), rt_device_count(ctx),
           (unsigned long long)n, (unsigned long long)st.camera_samples, (unsigned long long)st.rays_closest, (unsigned long long)st.rays_any);
    rt_destroy(ctx);

    f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 2; }
    fwrite(prim, 4, n, f); fwrite(t, 4, n, f); fwrite(bary, 4, 3 * n, f); fwrite(occ, 1, n, f);
    fwrite(rgb, 4, npix * 3, f); fwrite(w, 4, npix, f); fwrite(rgb8, 1, npix * 3, f);
    fclose(f);
    return 0;
}
