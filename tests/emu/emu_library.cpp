// The whole device library on the CPU: this translation unit includes raytracing_amd/csrc/device/rt_device.hip unchanged
// (C ABI, host code, kernels) against the stand-ins tests/emu/hip/hip_runtime.h and tests/emu/rccl/rccl.h, and serves the
// C ABI of include/raytracing_amd.h over its standard input and output, one call per request. Built with
// -fsanitize=address,undefined, every "device" buffer is a heap block of its exact size, every release is a free, every launch runs
// its grid fibre by fibre before it returns: a stale pointer, a buffer one element short, a table index out of range or
// a barrier that part of a workgroup misses ends the process with a report on stderr. tests/emu_context.py is the client
// (an object with the methods of raytracing_amd.device.Context); the tests compare what comes back with the oracle.
// A checker for the library's source, not a second way to run it: the library itself has no CPU path.
//
// Request:  uint32 op, uint64 payload bytes, payload.      Response: int32 rc, uint64 payload bytes, payload
// (rc != 0: the payload is rt_last_error's text). Ops and payloads (all little-endian, arrays packed):
//   1 create    int32 n, int32 device_ids[n]
//   2 upload    uint32 n_nodes, n_leaf_idx, n_prims, n_materials, n_lights, n_spheres, flags (1 tri_n, 2 tri_uv, 4 shape_type);
//               float bounds[6]; nodes, leaf_idx, tri_p, [tri_n], [tri_uv], material_id, light_id, materials, lights,
//               [shape_type], spheres
//   3 camera    float raster_to_camera[16], camera_to_world[16]; int32 width, height, sample_bounds[4]
//   4 film      float radius_x, radius_y; int32 cropped_bounds[4]; float max_sample_luminance
//   5 option    int32 option, value
//   6 closest   uint64 n; uint32 has_tmax; float o[3n], d[3n], [tmax[n]]          -> int32 prim[n], float t[n], bary[3n]
//   7 any       as 6                                                              -> uint8 occluded[n]
//   8 render    rt_render_params; uint32 entry (0 rt_render, 1 rt_render_noalloc_dev); float rgb[3 W H], w[W H]  -> the same two
//   9 develop   float film_scale; float rgb[3 W H], w[W H]                        -> uint8 rgb8[3 W H]
//  10 stats                                                                       -> rt_stats
//  11 reset_stats
//  12 destroy
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../raytracing_amd/csrc/device/rt_device.hip"

namespace {

struct Reader {
    const std::vector<unsigned char> &buf;
    size_t at = 0;
    template <typename T> T one() {
        T v;
        take(&v, sizeof(T));
        return v;
    }
    template <typename T> std::vector<T> many(size_t count) {
        std::vector<T> v(count);   // exact size: the library must not read past what the caller owns
        take(v.data(), count * sizeof(T));
        return v;
    }
    void take(void *dst, size_t bytes) {
        if (at + bytes > buf.size()) {
            fprintf(stderr, "emu_library: request shorter than its layout\n");
            exit(2);
        }
        if (bytes) memcpy(dst, buf.data() + at, bytes);
        at += bytes;
    }
};

bool read_exact(void *dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, stdin) == bytes; }

void respond(int rc, const void *payload, size_t bytes) {
    const int32_t r = rc;
    const uint64_t n = bytes;
    fwrite(&r, 4, 1, stdout);
    fwrite(&n, 8, 1, stdout);
    if (bytes) fwrite(payload, 1, bytes, stdout);
    fflush(stdout);
}

template <typename T> void append(std::vector<unsigned char> &out, const std::vector<T> &v) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(v.data());
    out.insert(out.end(), p, p + v.size() * sizeof(T));
}

// One function per request (a single function with every request's buffers takes the sanitizer minutes to compile).
struct Session {
    rt_context *ctx = nullptr;
    int width = 0, height = 0;
};

int op_create(Session &s, Reader &rd) {
    const int32_t n = rd.one<int32_t>();
    const std::vector<int32_t> ids = rd.many<int32_t>((size_t)n);
    return rt_create_multi(ids.data(), n, &s.ctx);
}

int op_upload(Session &s, Reader &rd) {
    rt_scene_desc sd = {};
    sd.n_nodes = rd.one<uint32_t>(); sd.n_leaf_idx = rd.one<uint32_t>(); sd.n_prims = rd.one<uint32_t>();
    sd.n_materials = rd.one<uint32_t>(); sd.n_lights = rd.one<uint32_t>(); sd.n_spheres = rd.one<uint32_t>();
    const uint32_t flags = rd.one<uint32_t>();
    rd.take(sd.bounds, sizeof sd.bounds);
    const auto nodes = rd.many<uint32_t>(2 * (size_t)sd.n_nodes);
    const auto leafIdx = rd.many<int32_t>(sd.n_leaf_idx);
    const auto triP = rd.many<float>(9 * (size_t)sd.n_prims);
    const auto triN = rd.many<float>((flags & 1u) ? 9 * (size_t)sd.n_prims : 0);
    const auto triUv = rd.many<float>((flags & 2u) ? 6 * (size_t)sd.n_prims : 0);
    const auto matId = rd.many<int32_t>(sd.n_prims), lightId = rd.many<int32_t>(sd.n_prims);
    const auto mats = rd.many<rt_material>(sd.n_materials);
    const auto lights = rd.many<rt_light>(sd.n_lights);
    const auto shapeType = rd.many<uint8_t>((flags & 4u) ? sd.n_prims : 0);
    const auto spheres = rd.many<rt_sphere>(sd.n_spheres);
    sd.nodes = nodes.data();
    sd.leaf_idx = leafIdx.empty() ? nullptr : leafIdx.data();
    sd.tri_p = triP.data();
    sd.tri_n = triN.empty() ? nullptr : triN.data();
    sd.tri_uv = triUv.empty() ? nullptr : triUv.data();
    sd.material_id = matId.data();
    sd.light_id = lightId.data();
    sd.materials = mats.data();
    sd.lights = lights.empty() ? nullptr : lights.data();
    sd.shape_type = shapeType.empty() ? nullptr : shapeType.data();
    sd.spheres = spheres.empty() ? nullptr : spheres.data();
    return rt_upload_scene(s.ctx, &sd);   // the arrays above are released right after: the library keeps copies
}

int op_camera(Session &s, Reader &rd) {
    float r2c[16], c2w[16];
    int32_t v[6];
    rd.take(r2c, sizeof r2c); rd.take(c2w, sizeof c2w); rd.take(v, sizeof v);
    const int rc = rt_set_camera(s.ctx, r2c, c2w, v[0], v[1], &v[2]);
    if (rc == RT_OK) { s.width = v[0]; s.height = v[1]; }
    return rc;
}

int op_film(Session &s, Reader &rd) {
    const float rx = rd.one<float>(), ry = rd.one<float>();
    int32_t cb[4];
    rd.take(cb, sizeof cb);
    return rt_set_film(s.ctx, rx, ry, cb, rd.one<float>());
}

int op_option(Session &s, Reader &rd) {
    const int32_t option = rd.one<int32_t>(), value = rd.one<int32_t>();
    return rt_set_option(s.ctx, option, value);
}

int op_trace(Session &s, Reader &rd, bool closest, std::vector<unsigned char> &out) {
    const uint64_t n = rd.one<uint64_t>();
    const uint32_t hasTMax = rd.one<uint32_t>();
    const auto o = rd.many<float>(3 * n), d = rd.many<float>(3 * n);
    const auto tmax = rd.many<float>(hasTMax ? n : 0);
    int rc;
    if (closest) {
        std::vector<int32_t> prim(n);
        std::vector<float> t(n), bary(3 * n);
        rc = rt_trace_closest(s.ctx, n, o.data(), d.data(), hasTMax ? tmax.data() : nullptr, prim.data(), t.data(), bary.data());
        append(out, prim); append(out, t); append(out, bary);
    } else {
        std::vector<uint8_t> occ(n);
        rc = rt_trace_any(s.ctx, n, o.data(), d.data(), hasTMax ? tmax.data() : nullptr, occ.data());
        append(out, occ);
    }
    return rc;
}

int op_render(Session &s, Reader &rd, std::vector<unsigned char> &out) {
    const rt_render_params p = rd.one<rt_render_params>();
    const uint32_t entry = rd.one<uint32_t>();
    const size_t npix = (size_t)s.width * s.height;
    auto rgb = rd.many<float>(3 * npix), w = rd.many<float>(npix);
    int rc;
    if (entry == 0) {
        rc = rt_render(s.ctx, &p, rgb.data(), w.data());
    } else {
        rc = rt_render_noalloc_dev(s.ctx, &p, rgb.data(), w.data());   // "device" pointers are heap pointers here
        if (rc == RT_OK) rc = rt_synchronize(s.ctx);
    }
    append(out, rgb); append(out, w);
    return rc;
}

int op_develop(Session &s, Reader &rd, std::vector<unsigned char> &out) {
    const float scale = rd.one<float>();
    const size_t npix = (size_t)s.width * s.height;
    const auto rgb = rd.many<float>(3 * npix), w = rd.many<float>(npix);
    std::vector<uint8_t> rgb8(3 * npix);
    const int rc = rt_develop(s.ctx, rgb.data(), w.data(), scale, rgb8.data());
    append(out, rgb8);
    return rc;
}

int op_stats(Session &s, std::vector<unsigned char> &out) {
    rt_stats st = {};
    const int rc = rt_get_stats(s.ctx, &st);
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&st);
    out.assign(p, p + sizeof st);
    return rc;
}

}   // namespace

int main() {
    Session s;
    for (;;) {
        uint32_t op;
        uint64_t bytes;
        if (!read_exact(&op, 4) || !read_exact(&bytes, 8)) break;   // the client closed the pipe
        std::vector<unsigned char> payload(bytes);
        if (!read_exact(payload.data(), bytes)) return 2;
        Reader rd{payload};
        std::vector<unsigned char> out;
        int rc = RT_OK;
        switch (op) {
            case 1: rc = op_create(s, rd); break;
            case 2: rc = op_upload(s, rd); break;
            case 3: rc = op_camera(s, rd); break;
            case 4: rc = op_film(s, rd); break;
            case 5: rc = op_option(s, rd); break;
            case 6: rc = op_trace(s, rd, true, out); break;
            case 7: rc = op_trace(s, rd, false, out); break;
            case 8: rc = op_render(s, rd, out); break;
            case 9: rc = op_develop(s, rd, out); break;
            case 10: rc = op_stats(s, out); break;
            case 11: rc = rt_reset_stats(s.ctx); break;
            case 12:
                rt_destroy(s.ctx);
                s.ctx = nullptr;
                break;
            default: fprintf(stderr, "emu_library: unknown op %u\n", op); return 2;
        }
        if (rc != RT_OK) {
            const std::string msg = rt_last_error(s.ctx);
            respond(rc, msg.data(), msg.size());
        } else {
            respond(RT_OK, out.data(), out.size());
        }
    }
    if (s.ctx) rt_destroy(s.ctx);
    return 0;
}
